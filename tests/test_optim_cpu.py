"""hulc2_amd.optim.Adam away from a parameter arena (CPU, plain parameters): it is torch.optim.Adam — same values bit for bit, same state_dict,
checkpoints interchange (reference: hulc2/models/hulc2.py:185-198 instantiates `optimizer._target_`)."""
import copy
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_fallback_is_torch_adam():
    from hulc2_amd.optim import Adam

    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(7, 5, generator=g), torch.randn(5, generator=g), torch.randn(3, generator=g)]
    a = [torch.nn.Parameter(w.clone()) for w in ws]
    b = [torch.nn.Parameter(w.clone()) for w in ws]
    oa, ob = Adam(a, lr=2e-4, weight_decay=0.01), torch.optim.Adam(b, lr=2e-4, weight_decay=0.01)
    for step in range(4):
        for pa, pb in zip(a[:2], b[:2]):                          # (the third parameter never gets a gradient: no state, no update)
            gr = torch.randn(pa.shape, generator=g)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        if step == 1:                                             # checkpoints cross over mid-run
            sa, sb = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
            oa.load_state_dict(sb); ob.load_state_dict(sa)
    assert oa.fused_launches == 0
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys() == {0, 1}
    for k in sa["state"]:
        for key in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(torch.as_tensor(sa["state"][k][key]), torch.as_tensor(sb["state"][k][key]))
