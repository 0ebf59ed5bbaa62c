#!/bin/bash
# usage: tools/isa_waits.sh <file.hip> <kernel-name-substring> : order of vector loads / vmcnt waits / MFMAs / barriers in one kernel's ISA
# (how the "no vector-memory wait in front of the MFMA loop" rule of DESIGN.md section 3 is checked)
set -e
T=$(mktemp -d); S=$T/k.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I include -I hulc2_amd/csrc -S --cuda-device-only "$1" -o $S 2>/dev/null
a=$(grep -n "^_Z[A-Za-z0-9_]*$2[A-Za-z0-9_]*:" $S | head -1 | cut -d: -f1)
[ -z "$a" ] && { echo "no kernel matches $2"; grep -o "^_Z[A-Za-z0-9_]*:" $S | head -20; exit 1; }
b=$(awk -v a=$a 'NR>a && /s_endpgm/{print NR; exit}' $S)
echo "kernel: $(sed -n "${a}p" $S)"
sed -n "${a},${b}p" $S | grep -n "global_load\|buffer_load\|s_waitcnt vmcnt\|v_mfma\|s_barrier\|global_store\|scratch_\|global_atomic" | awk '{print $2" "$3}' | sed -E 's/v\[[0-9:]+\],?//; s/v[0-9]+,?//' | uniq -c
rm -rf $T
