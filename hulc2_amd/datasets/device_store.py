"""DeviceEpisodeStore — the reference's shared-memory play dataset, resident in HBM instead of host shared memory.

Reference behaviour mirrored here (same names, same argument meaning):
  * hulc2/datasets/utils/shared_memory_loader.py:59-77   valid window starts: every frame of an episode that still has
                                                          `min_window_size` frames after it; `episode_counters` = step inside
                                                          the episode
  * hulc2/datasets/shm_dataset.py:77-100                  get_window_size: longest window that stays inside the episode, then
                                                          uniform in [min, max] (training) or a hash of the index (validation)
  * hulc2/datasets/base_dataset.py:26-28                  get_validation_window_size (pyhash.fnv1_32, seed 0)
  * hulc2/datasets/base_dataset.py:94-165                 __getitem__ + pad_sequence: frames / observations repeat the last step,
                                                          relative actions zero-pad dims 0..5 and repeat the gripper dim
  * hulc2/utils/transforms.py:85-106                      RandomShiftsAug draws one integer shift per frame

MI355X design: an MI355X holds 288 GB, CALVIN's task_D split is ~70 GB of uint8 frames — the whole split stays in HBM as it is
stored on disk (uint8 NHWC).  A batch is never materialised: a window is a row of store frame numbers (padding = a repeated
number) that conv1 follows while staging (hulc_conv_desc.frame_index), and the per-frame augmentation shift rides along.  Only
the small vectors (actions, proprioception) are gathered, by hulc_window_rows.  All outputs live in fixed device buffers so a
captured training graph can be replayed after `batch()` refreshed them."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .. import kernels as kn


def fnv1_32(data: bytes, seed: int = 0) -> int:
    """pyhash.fnv1_32 (requirements.txt:12, no version pin): FNV-1, 32 bit, with the hash value starting at `seed` (default 0)"""
    h = seed & 0xFFFFFFFF
    for c in data:
        h = (h * 16777619) & 0xFFFFFFFF
        h ^= c
    return h


def validation_window_size(idx: int, min_window_size: int, max_window_size: int) -> int:
    """base_dataset.py:26-28"""
    return min_window_size + fnv1_32(str(idx).encode()) % (max_window_size - min_window_size + 1)


class DeviceEpisodeStore:
    AUG_PAD = {"rgb_static": 10, "rgb_gripper": 4}     # conf/datamodule/transforms/rand_shift.yaml:5,12

    def __init__(self, rgb: Dict[str, torch.Tensor], rel_actions: torch.Tensor, robot_obs: torch.Tensor, ep_start_end_ids,
                 min_window_size: int = 20, max_window_size: int = 32, pad: bool = True, validation: bool = False,
                 scene_obs: Optional[torch.Tensor] = None, lang_emb: Optional[torch.Tensor] = None,
                 lang_lookup: Optional[Sequence[int]] = None, device=None, seed: int = 0, aux_lang_loss_window: int = 1):
        """rgb: {"rgb_static": (n, 200, 200, 3) uint8, "rgb_gripper": (n, 84, 84, 3) uint8} — the frames of all episodes back to
        back; rel_actions (n, 7), robot_obs (n, 15) [, scene_obs (n, 24)] fp32; ep_start_end_ids (E, 2): first / last store frame of
        each episode (inclusive, as ep_start_end_ids.npy).  lang_emb (n_ann, 384) + lang_lookup (one annotation per window start)
        make it the language dataset."""
        if min_window_size > max_window_size:
            raise ValueError(f"min_window_size {min_window_size} > max_window_size {max_window_size}")   # base_dataset.py:103-105
        if not pad and min_window_size != max_window_size:
            raise NotImplementedError("varying window sizes need pad=True (fixed step shapes)")
        dev = torch.device(device) if device is not None else next(iter(rgb.values())).device
        n = rel_actions.shape[0]
        self.rgb = {}
        for k, v in rgb.items():
            if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[-1] != 3 or v.shape[0] != n:
                raise TypeError(f"{k}: frames are stored uint8 (n, H, W, 3)")
            self.rgb[k] = v.to(dev).contiguous()
        self.rel_actions = rel_actions.to(dev, torch.float32).contiguous()
        self.robot_obs = robot_obs.to(dev, torch.float32).contiguous()
        self.scene_obs = None if scene_obs is None else scene_obs.to(dev, torch.float32).contiguous()
        self.lang_emb = None if lang_emb is None else lang_emb.to(dev, torch.float32).contiguous()
        self.device = dev
        self.min_window_size, self.max_window_size, self.pad, self.validation = min_window_size, max_window_size, pad, validation
        lookup, counters = [], []
        for a, b in np.asarray(ep_start_end_ids, dtype=np.int64).reshape(-1, 2):
            if not (0 <= a <= b < n):
                raise ValueError(f"episode ({a}, {b}) outside the store of {n} frames")
            for j, f in enumerate(range(a, b + 1 - min_window_size)):
                lookup.append(f)
                counters.append(j)
        self.episode_lookup = np.asarray(lookup, dtype=np.int64)       # store frame of each valid window start
        self.episode_counters = np.asarray(counters, dtype=np.int64)
        self.lang_lookup = None if lang_lookup is None else np.asarray(lang_lookup, dtype=np.int64)
        if self.lang_lookup is not None and len(self.lang_lookup) != len(self.episode_lookup):
            raise ValueError("lang_lookup holds one annotation number per window start")
        self.aux_lang_loss_window = aux_lang_loss_window
        self.np_rng = np.random.RandomState(seed)
        self.generator = torch.Generator().manual_seed(seed)
        self._buf: Dict[int, Dict] = {}

    @classmethod
    def from_language_annotations(cls, rgb, rel_actions, robot_obs, lang_data: Dict, frame0: int = 0, load_lang_embeddings: bool = True,
                                  min_window_size: int = 20, max_window_size: int = 32, aux_lang_loss_window: int = 8, skip_frames: int = 1,
                                  **kw) -> "DeviceEpisodeStore":
        """The language dataset (file-format half): `lang_data` is the dict stored in
        `<split>/lang_paraphrase-MiniLM-L3-v2/auto_lang_ann.npy` — {"language": {"ann": [...], "task": [...], "emb": (n, 1, 384)},
        "info": {"indx": [(first, last), ...]}} with dataset frame numbers; `frame0` = dataset frame number of store frame 0.
        Window starts / annotation lookup as hulc2/datasets/utils/shared_memory_loader.py:133-140 and npz_dataset.py:182-192 (every frame
        of an annotated span that still has `min_window_size` frames after it), `use_for_aux_lang_loss` as shm_dataset.py:150-158,
        conf/datamodule/datasets/lang_dataset/lang_shm.yaml:5-6,12 for the defaults.  Only the precomputed embeddings are supported
        (`load_lang_embeddings: true`): sentences would need the SBERT encoder."""
        if not load_lang_embeddings:
            raise NotImplementedError("sentences need the SBERT encoder: set datamodule.datasets.lang_dataset.load_lang_embeddings=true "
                                      "(the precomputed 'emb' entries of auto_lang_ann.npy are used)")
        emb = torch.as_tensor(np.asarray(lang_data["language"]["emb"], dtype=np.float32))
        emb = emb.reshape(emb.shape[0], -1)                              # (n, 1, 384) -> (n, 384): process_language squeezes
        spans = [(int(a) - frame0, int(b) - frame0) for a, b in lang_data["info"]["indx"]]
        self = cls(rgb, rel_actions, robot_obs, spans, min_window_size, max_window_size, lang_emb=emb,
                   lang_lookup=None, aux_lang_loss_window=aux_lang_loss_window, **kw)
        lookup, lang_lookup = [], []
        for i, (a, b) in enumerate(spans):
            for cnt, f in enumerate(range(a, b + 1 - min_window_size)):
                if cnt % skip_frames == 0:
                    lookup.append(f)
                    lang_lookup.append(i)
        self.episode_lookup = np.asarray(lookup, dtype=np.int64)
        self.episode_counters = self.episode_lookup.copy()               # consecutive frames of one span <=> consecutive counters
        self.lang_lookup = np.asarray(lang_lookup, dtype=np.int64)
        return self

    def use_for_aux_lang_loss(self, idx: int) -> bool:
        """shm_dataset.py:150-158: true for the last `aux_lang_loss_window` window starts of an annotation"""
        w = self.aux_lang_loss_window
        return bool(idx + w < len(self.lang_lookup) and self.lang_lookup[idx] < self.lang_lookup[idx + w])

    def __len__(self) -> int:
        return len(self.episode_lookup)

    # ---- shm_dataset.py:77-100 ---------------------------------------------------------------------------------------
    def max_window(self, idx: int) -> int:
        cnt, lo, hi = self.episode_counters, self.min_window_size, self.max_window_size
        diff = hi - lo
        if len(cnt) <= idx + diff:                                     # last episode
            return lo + len(cnt) - idx - 1
        if cnt[idx + diff] != cnt[idx] + diff:                         # fewer than `diff` more starts before the next episode
            run = cnt[idx:idx + diff + 1] - (cnt[idx] + np.arange(diff + 1))
            return min(hi, int(lo + np.nonzero(run)[0][0] - 1))
        return hi

    def get_window_size(self, idx: int) -> int:
        if self.min_window_size == self.max_window_size:               # base_dataset.py:98-99
            return self.max_window_size
        mw = self.max_window(idx)
        if self.validation:
            return validation_window_size(idx, self.min_window_size, mw)
        return int(self.np_rng.randint(self.min_window_size, mw + 1))

    # ---- base_dataset.py:94-147 on device ------------------------------------------------------------------------------
    def _buffers(self, B: int) -> Dict:
        """Fixed device buffers of a batch of B windows.  Everything the host decides per step (window starts and sizes, dataset indices,
        shift draws, annotation numbers, aux-loss flags) lives in ONE int32 block: it is filled in a pinned host mirror (two slots, so a
        slot is never rewritten while its copy may still be in flight) and moved with one asynchronous copy — a pageable `copy_` per field
        made the host wait for the previous step's GPU work every time."""
        if B not in self._buf:
            S, dev = self.max_window_size, self.device
            keys = sorted(self.rgb)
            off, n = {}, 0
            for name, cnt in [("idx64", 2 * B), ("starts", B), ("sizes", B), ("ann", B), ("use", B)] + [(k + "_shift", B * S * 2) for k in keys]:
                off[name] = (n, cnt); n += cnt
            blk = torch.zeros(n, dtype=torch.int32, device=dev)
            view = lambda t, name: t[off[name][0]:off[name][0] + off[name][1]]
            b = {"block": blk, "off": off, "slot": 0, "events": [None, None],
                 "host": [torch.zeros(n, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(n, dtype=torch.int32) for _ in range(2)],
                 "starts": view(blk, "starts"), "sizes": view(blk, "sizes"), "idx": view(blk, "idx64").view(torch.int64),
                 "ann": view(blk, "ann"), "use": view(blk, "use"),
                 "index": torch.zeros(B, S, dtype=torch.int32, device=dev),
                 "actions": torch.zeros(B, S, self.rel_actions.shape[1], device=dev),
                 "robot_obs": torch.zeros(B, S, self.robot_obs.shape[1], device=dev)}
            for k in keys:
                b[k + "_shift"] = view(blk, k + "_shift").view(B, S, 2)
            if self.scene_obs is not None:
                b["scene_obs"] = torch.zeros(B, S, self.scene_obs.shape[1], device=dev)
            if self.lang_emb is not None:
                b["lang"] = torch.zeros(B, self.lang_emb.shape[1], device=dev)
                b["use_aux"] = torch.zeros(B, dtype=torch.bool, device=dev)
            self._buf[B] = b
        return self._buf[B]

    def batch(self, idxs: Sequence[int], window_sizes: Optional[Sequence[int]] = None, shifts: Optional[Dict[str, torch.Tensor]] = None) -> Dict:
        """One modality's batch dict for Hulc2.training_step / validation_step from the window starts `idxs` (dataset indices as the
        reference's sampler hands them to __getitem__).  window_sizes / shifts override the draws (parity tests)."""
        B, S = len(idxs), self.max_window_size
        idxs = np.asarray(idxs, dtype=np.int64)
        sizes = np.asarray([self.get_window_size(int(i)) for i in idxs] if window_sizes is None else window_sizes, dtype=np.int32)
        if sizes.min() < 1 or sizes.max() > S:
            raise ValueError("window sizes must lie in [1, max_window_size]")
        buf = self._buffers(B)
        slot = buf["slot"]; buf["slot"] = slot ^ 1
        if buf["events"][slot] is not None:
            buf["events"][slot].synchronize()                         # the copy that last read this host slot has finished (two steps ago)
        host, off = buf["host"][slot], buf["off"]
        hv = lambda name: host[off[name][0]:off[name][0] + off[name][1]]
        hv("idx64").view(torch.int64).copy_(torch.from_numpy(idxs))
        hv("starts").copy_(torch.from_numpy(self.episode_lookup[idxs].astype(np.int32)))
        hv("sizes").copy_(torch.from_numpy(sizes))
        with_shift = shifts is not None or not self.validation         # the validation transforms carry no RandomShiftsAug
        if with_shift:
            for k in self.rgb:
                pad = self.AUG_PAD.get(k, 0)
                if shifts is not None:
                    hv(k + "_shift").copy_(shifts[k].reshape(-1).to(torch.int32))
                else:
                    torch.randint(0, 2 * pad + 1, (B * S * 2,), generator=self.generator, dtype=torch.int32, out=hv(k + "_shift"))
        if self.lang_emb is not None:
            hv("ann").copy_(torch.from_numpy(self.lang_lookup[idxs].astype(np.int32)))
            hv("use").copy_(torch.from_numpy(np.asarray([self.use_for_aux_lang_loss(int(i)) for i in idxs], dtype=np.int32)))
        buf["block"].copy_(host, non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event(); ev.record(); buf["events"][slot] = ev
        kn.window_index(buf["starts"], buf["sizes"], B, S, buf["index"])
        kn.window_rows(self.rel_actions, buf["starts"], buf["sizes"], B, S, buf["actions"], zero_cols=(0, self.rel_actions.shape[1] - 1))
        kn.window_rows(self.robot_obs, buf["starts"], buf["sizes"], B, S, buf["robot_obs"])
        state_info = {"robot_obs": buf["robot_obs"]}
        if self.scene_obs is not None:
            kn.window_rows(self.scene_obs, buf["starts"], buf["sizes"], B, S, buf["scene_obs"])
            state_info["scene_obs"] = buf["scene_obs"]
        rgb_obs = {}
        for k, store in self.rgb.items():
            rgb_obs[k] = store
            rgb_obs[k + "_index"] = buf["index"]
            if with_shift:
                rgb_obs[k + "_shift"] = buf[k + "_shift"]
        out = {"rgb_obs": rgb_obs, "depth_obs": {}, "robot_obs": buf["robot_obs"], "actions": buf["actions"], "state_info": state_info,
               "idx": buf["idx"], "window_sizes": buf["sizes"]}
        if self.lang_emb is not None:
            torch.index_select(self.lang_emb, 0, buf["ann"], out=buf["lang"])
            torch.ne(buf["use"], 0, out=buf["use_aux"])
            out["lang"] = buf["lang"]
            out["use_for_aux_lang_loss"] = buf["use_aux"]
        return out
