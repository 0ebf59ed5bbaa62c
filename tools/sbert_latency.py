"""latency of SBertLang.encode_tokens: 32 sentences x 16 tokens, the language batch of a training step"""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.models.language_encoders import SBertLang
from tests.test_oracle_golden import _bert_sd
dev = torch.device('cuda')
kn.set_compute("bf16")
enc = SBertLang().to(dev); enc.load_bert_state_dict(_bert_sd(1))
for B, S in ((32, 16), (1, 12), (64, 32)):
    ids = torch.randint(0, 30522, (B, S)).to(dev); mask = torch.ones(B, S, dtype=torch.long, device=dev)
    for _ in range(3): enc.encode_tokens(ids, mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): enc.encode_tokens(ids, mask)
    e1.record(); torch.cuda.synchronize()
    print(f"SBertLang.encode_tokens B={B} S={S}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us")
