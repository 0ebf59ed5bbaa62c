"""CPU ORACLE (test infrastructure only) for the HULC++ low-level policy training_step.

This file is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it.  It restates, in plain fp32 torch functional ops over a flat
state_dict that uses the reference's parameter names, what the reference computes on the path
`Hulc2.training_step` (/root/reference/hulc2/models/hulc2.py:336-442).  It never touches a GPU kernel.

Pinning: every function below is checked in tests/test_oracle_golden.py against fixtures under
tests/golden/ that were produced by importing the reference's own leaf modules in the build container
(oracle/gen_golden.py).  Exceptions — `world_to_tcp_frame` follows pytorch3d (un-vendored, version
unpinned in the reference's requirements.txt:23): PARITY UNPINNED, checked by properties only.
SBERT (sentence-transformers, unpinned) and R3M (empty submodule) are restated from their public definitions
(`minilm_sentence_embedding`, `r3m_trunk_features`): PARITY UNPINNED against the reference's third-party code; MiniLM is
pinned against transformers' BertModel, the ResNet trunk against torch's own nn layers (tests/test_oracle_golden.py).

Each function cites the reference lines it follows.
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------
# perceptual encoders
# ------------------------------------------------------------------------------------------------
def spatial_softmax(x: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """hulc2/models/perceptual_encoders/vision_network.py:74-108.

    x_map varies along rows (meshgrid(linspace(num_cols), linspace(num_rows), indexing="ij") flattened
    row-major), y_map along columns; output interleaves (ex, ey) per channel.
    """
    n, c, h, w = x.shape
    # reference builds the maps from (num_cols=h_out?, num_rows) = the conv output size; both are
    # square on this path.  grid_x[i, j] = lin_cols[i], grid_y[i, j] = lin_rows[j].
    lin_a = torch.linspace(-1.0, 1.0, w, dtype=x.dtype)
    lin_b = torch.linspace(-1.0, 1.0, h, dtype=x.dtype)
    x_map = lin_a.reshape(-1, 1).expand(w, h).reshape(-1)
    y_map = lin_b.reshape(1, -1).expand(w, h).reshape(-1)
    flat = x.contiguous().view(-1, h * w)
    att = F.softmax(flat / temperature, dim=1)
    ex = torch.sum(x_map * att, dim=1, keepdim=True)
    ey = torch.sum(y_map * att, dim=1, keepdim=True)
    return torch.cat((ex, ey), 1).view(-1, c * 2)


def _conv_stack(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """The three un-padded convs shared by both cameras: 8x8 s4 -> 4x4 s2 -> 3x3 s1, ReLU after each
    (vision_network.py:36-47, vision_network_gripper.py:11-20)."""
    x = F.relu(F.conv2d(x, sd[p + "conv_model.0.weight"], sd[p + "conv_model.0.bias"], stride=4))
    x = F.relu(F.conv2d(x, sd[p + "conv_model.2.weight"], sd[p + "conv_model.2.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[p + "conv_model.4.weight"], sd[p + "conv_model.4.bias"], stride=1))
    return x


def _fc_tail(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """fc1 (+ReLU, dropout p=0) -> fc2 -> LayerNorm (vision_network.py:49-53,60-64)."""
    x = F.relu(F.linear(x, sd[p + "fc1.0.weight"], sd[p + "fc1.0.bias"]))
    x = F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    return F.layer_norm(x, (x.shape[-1],), sd[p + "ln.weight"], sd[p + "ln.bias"], 1e-5)


def vision_network_static(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """VisionNetwork.forward, hulc2/models/perceptual_encoders/vision_network.py:55-65."""
    x = _conv_stack(sd, p, x)
    x = spatial_softmax(x, float(sd[p + "spatial_softmax.temperature"]) if p + "spatial_softmax.temperature" in sd else 1.0)
    return _fc_tail(sd, p, x)


def vision_network_gripper(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nature_cnn + VisionNetwork.forward, vision_network_gripper.py:11-26,82-89."""
    x = _conv_stack(sd, p, x)
    x = torch.flatten(x, 1)
    x = F.relu(F.linear(x, sd[p + "conv_model.7.weight"], sd[p + "conv_model.7.bias"]))
    return _fc_tail(sd, p, x)


def concat_encoders(sd: SD, p: str, rgb_static: torch.Tensor, rgb_gripper: torch.Tensor) -> torch.Tensor:
    """ConcatEncoders.forward with proprio: none (concat_encoders.py:59-109)."""
    b, s, c, h, w = rgb_static.shape
    if p + "rgb_static_encoder.r3m.convnet.conv1.weight" in sd:      # rgb_static: r3m (vision_r3m.py:24-32), frames in [0, 255]
        e1 = vision_r3m(sd, p + "rgb_static_encoder.", rgb_static.reshape(-1, c, h, w)).reshape(b, s, -1)
    else:
        e1 = vision_network_static(sd, p + "rgb_static_encoder.", rgb_static.reshape(-1, c, h, w)).reshape(b, s, -1)
    b, s, c, h, w = rgb_gripper.shape
    e2 = vision_network_gripper(sd, p + "rgb_gripper_encoder.", rgb_gripper.reshape(-1, c, h, w)).reshape(b, s, -1)
    return torch.cat([e1, e2], dim=-1)


# ------------------------------------------------------------------------------------------------
# goal encoders, plan proposal
# ------------------------------------------------------------------------------------------------
def _mlp3_ln(sd: SD, p: str, x: torch.Tensor, i0: int) -> torch.Tensor:
    x = F.relu(F.linear(x, sd[f"{p}mlp.{i0}.weight"], sd[f"{p}mlp.{i0}.bias"]))
    x = F.relu(F.linear(x, sd[f"{p}mlp.{i0 + 2}.weight"], sd[f"{p}mlp.{i0 + 2}.bias"]))
    x = F.linear(x, sd[f"{p}mlp.{i0 + 4}.weight"], sd[f"{p}mlp.{i0 + 4}.bias"])
    return F.layer_norm(x, (x.shape[-1],), sd[p + "ln.weight"], sd[p + "ln.bias"], 1e-5)


def visual_goal_encoder(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """VisualGoalEncoder.forward, hulc2/models/encoders/goal_encoders.py:29-34 (mlp indices 0,2,4)."""
    return _mlp3_ln(sd, p, x, 0)


def language_goal_encoder(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """LanguageGoalEncoder.forward with lang_net=None and word dropout 0, goal_encoders.py:62-71
    (the Dropout sits at index 0 so the Linear indices are 1,3,5)."""
    return _mlp3_ln(sd, p, x, 1)


def plan_proposal(sd: SD, p: str, emb0: torch.Tensor, goal: torch.Tensor) -> torch.Tensor:
    """PlanProposalNetwork.forward -> prior logits, plan_proposal_net.py:42-47."""
    x = torch.cat([emb0, goal], dim=-1)
    for i in (0, 2, 4, 6):
        x = F.relu(F.linear(x, sd[f"{p}fc_model.{i}.weight"], sd[f"{p}fc_model.{i}.bias"]))
    return F.linear(x, sd[p + "fc_state.0.weight"], sd[p + "fc_state.0.bias"])


# ------------------------------------------------------------------------------------------------
# plan recognition transformer (post-norm nn.TransformerEncoderLayer written out)
# ------------------------------------------------------------------------------------------------
def _encoder_layer(sd: SD, p: str, x: torch.Tensor, nhead: int) -> torch.Tensor:
    """x: (S, B, E).  torch.nn.TransformerEncoderLayer defaults as the reference instantiates it
    (plan_recognition_net.py:115-117): post-norm, ReLU, eps 1e-5, packed in_proj; dropout off."""
    S, B, E = x.shape
    dh = E // nhead
    qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)

    def heads(t):  # (S, B, E) -> (B*h, S, dh), feature e = head*dh + d
        return t.contiguous().view(S, B * nhead, dh).transpose(0, 1)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(dh), dim=-1)
    o = torch.bmm(att, v).transpose(0, 1).contiguous().view(S, B, E)
    o = F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + o, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    ff = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                  sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return F.layer_norm(x + ff, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)


def plan_recognition(sd: SD, p: str, emb: torch.Tensor, nhead: int = 8, num_layers: int = 2) -> Tuple[torch.Tensor, torch.Tensor]:
    """PlanRecognitionTransformersNetwork.forward (position_embedding=True, no normalisation flags,
    dropout disabled), plan_recognition_net.py:125-148.  Returns (posterior logits, seq_feat)."""
    B, S, E = emb.shape
    pos_ids = torch.arange(S, dtype=torch.long)          # integer indexing: bit-exact by construction
    x = emb + sd[p + "position_embeddings.weight"][pos_ids].unsqueeze(0)
    x = x.permute(1, 0, 2)
    for l in range(num_layers):
        x = _encoder_layer(sd, f"{p}transformer_encoder.layers.{l}.", x, nhead)
    x = F.linear(x.permute(1, 0, 2), sd[p + "fc.weight"], sd[p + "fc.bias"])
    seq_feat = torch.mean(x, dim=1)
    logits = F.linear(seq_feat, sd[p + "fc_state.0.weight"], sd[p + "fc_state.0.bias"])
    return logits, seq_feat


# ------------------------------------------------------------------------------------------------
# discrete latent plan: straight-through sample and KL balancing
# ------------------------------------------------------------------------------------------------
def straight_through_sample(logits: torch.Tensor, idx: torch.Tensor, cat: int = 32, cls: int = 32) -> torch.Tensor:
    """Independent(OneHotCategoricalStraightThrough).rsample() with the sampled class indices injected
    (distributions.py:23-27, hulc2.py:235-237): onehot + probs - probs.detach(), flattened."""
    lg = logits.reshape(*logits.shape[:-1], cat, cls)
    probs = torch.softmax(lg, dim=-1)
    onehot = F.one_hot(idx, cls).to(probs.dtype)
    return (onehot + (probs - probs.detach())).flatten(-2, -1)


def _cat_kl(post_logits: torch.Tensor, prior_logits: torch.Tensor, cat: int, cls: int) -> torch.Tensor:
    """KL(Independent(OneHotCategorical(post), 1) || ...(prior)) per sample = sum over categories of
    sum_c p_c (log p_c - log q_c)  (torch.distributions.kl._kl_categorical_categorical)."""
    lp = torch.log_softmax(post_logits.reshape(-1, cat, cls), dim=-1)
    lq = torch.log_softmax(prior_logits.reshape(-1, cat, cls), dim=-1)
    return (lp.exp() * (lp - lq)).sum(-1).sum(-1)


def kl_loss(pp_logits: torch.Tensor, pr_logits: torch.Tensor, kl_beta: float, kl_balancing_mix: float,
            cat: int = 32, cls: int = 32) -> torch.Tensor:
    """Hulc2.compute_kl_loss, hulc2.py:444-466 (KL balancing: alpha*KL(sg(post)||prior) + (1-alpha)*KL(post||sg(prior)))."""
    lhs = _cat_kl(pr_logits.detach(), pp_logits, cat, cls).mean()
    rhs = _cat_kl(pr_logits, pp_logits.detach(), cat, cls).mean()
    return (kl_balancing_mix * lhs + (1 - kl_balancing_mix) * rhs) * kl_beta


# ------------------------------------------------------------------------------------------------
# action decoder: 2-layer ReLU RNN + discretised logistic mixture + gripper CE
# ------------------------------------------------------------------------------------------------
def relu_rnn(sd: SD, p: str, x: torch.Tensor, num_layers: int = 2, h0: Optional[torch.Tensor] = None, return_state: bool = False):
    """nn.RNN(nonlinearity='relu', batch_first=True); h0 (num_layers, B, H) or zeros (decoders/utils/rnn.py:5-14)."""
    B, S, _ = x.shape
    inp = x
    finals = []
    for l in range(num_layers):
        w_ih, w_hh = sd[f"{p}weight_ih_l{l}"], sd[f"{p}weight_hh_l{l}"]
        b = sd[f"{p}bias_ih_l{l}"] + sd[f"{p}bias_hh_l{l}"]
        h = torch.zeros(B, w_hh.shape[0], dtype=x.dtype) if h0 is None else h0[l]
        outs = []
        for t in range(S):
            h = F.relu(F.linear(inp[:, t], w_ih) + F.linear(h, w_hh) + b)
            outs.append(h)
        finals.append(h)
        inp = torch.stack(outs, dim=1)
    return (inp, torch.stack(finals)) if return_state else inp


def decoder_forward(sd: SD, p: str, plan: torch.Tensor, emb: torch.Tensor, goal: torch.Tensor,
                    emb_slice=(64, 128), n_mix: int = 10, log_scale_min: float = -7.0, h0: Optional[torch.Tensor] = None,
                    return_state: bool = False):
    """LogisticDecoderRNN.forward, logistic_decoder_rnn.py:257-284 (h0: the carried hidden state of `act`, :105-107)."""
    pe = emb[..., emb_slice[0]:emb_slice[1]]
    B, S = pe.shape[0], pe.shape[1]
    x = torch.cat([plan.unsqueeze(1).expand(-1, S, -1), pe, goal.unsqueeze(1).expand(-1, S, -1)], dim=-1)
    h, h_n = relu_rnn(sd, p + "rnn.", x, h0=h0, return_state=True)
    probs = F.linear(h, sd[p + "prob_fc.weight"], sd[p + "prob_fc.bias"])
    means = F.linear(h, sd[p + "mean_fc.weight"], sd[p + "mean_fc.bias"])
    log_scales = torch.clamp(F.linear(h, sd[p + "log_scale_fc.weight"], sd[p + "log_scale_fc.bias"]), min=log_scale_min)
    grip = F.linear(h, sd[p + "gripper_fc.weight"], sd[p + "gripper_fc.bias"])
    A = probs.shape[-1] // n_mix
    out = (probs.view(B, S, A, n_mix), log_scales.view(B, S, A, n_mix), means.view(B, S, A, n_mix), grip)
    return (*out, h_n) if return_state else out


def sample_actions(logit_probs, log_scales, means, grip, u_mix, u_inv, gripper_bounds=(-1.0, 1.0)):
    """LogisticDecoderRNN._sample, logistic_decoder_rnn.py:231-255, with the two torch.rand draws passed in:
    u_mix (B,S,A,n_mix) and u_inv (B,S,A) raw uniforms in [0,1).  Returns (actions (B,S,A+1), selected mixture index (B,S,A))."""
    r1, r2 = 1e-5, 1.0 - 1e-5
    temp = (r1 - r2) * u_mix + r2
    temp = logit_probs - torch.log(-torch.log(temp))
    argmax = torch.argmax(temp, -1)
    dist = torch.eye(means.shape[-1], dtype=means.dtype)[argmax]
    ls = (dist * log_scales).sum(dim=-1)
    mu = (dist * means).sum(dim=-1)
    scales = torch.exp(ls)
    u = (r1 - r2) * u_inv + r2
    actions = mu + scales * (torch.log(u) - torch.log(1.0 - u))
    cmd = torch.tensor(gripper_bounds, dtype=means.dtype)[grip.argmax(dim=-1)]
    return torch.cat([actions, cmd.unsqueeze(-1)], 2), argmax


def logistic_loss(logit_probs, log_scales, means, actions, act_min=-1.0, act_max=1.0, num_classes=10,
                  log_scale_min=-7.0) -> torch.Tensor:
    """LogisticDecoderRNN._logistic_loss, logistic_decoder_rnn.py:181-228 (+ log_sum_exp :19-24)."""
    log_scales = torch.clamp(log_scales, min=log_scale_min)
    a = actions.unsqueeze(-1) * torch.ones_like(means)
    centered = a - means
    inv_stdv = torch.exp(-log_scales)
    half = (act_max - act_min) / 2.0 / (num_classes - 1)
    plus_in = inv_stdv * (centered + half)
    cdf_plus = torch.sigmoid(plus_in)
    min_in = inv_stdv * (centered - half)
    cdf_min = torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv_stdv * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    cdf_delta = cdf_plus - cdf_min
    log_probs = torch.where(
        a < act_min + 1e-3, log_cdf_plus,
        torch.where(a > act_max - 1e-3, log_one_minus_cdf_min,
                    torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)),
                                log_pdf_mid - math.log((num_classes - 1) / 2))))
    log_probs = log_probs + F.log_softmax(logit_probs, dim=-1)
    m = log_probs.max(dim=-1).values
    lse = m + torch.log(torch.sum(torch.exp(log_probs - m.unsqueeze(-1)), dim=-1))
    return -torch.sum(lse, dim=-1).mean()


def decoder_loss(logit_probs, log_scales, means, grip, actions, gripper_alpha: float = 1.0) -> torch.Tensor:
    """LogisticDecoderRNN._loss with discrete_gripper, logistic_decoder_rnn.py:133-152."""
    l_mix = logistic_loss(logit_probs, log_scales, means, actions[:, :, :-1])
    g = actions[:, :, -1].clone()
    g[g == -1] = 0                                        # integer label remap: bit-exact by construction
    l_grip = F.cross_entropy(grip.reshape(-1, 2), g.reshape(-1).long())
    return l_mix + gripper_alpha * l_grip


# ---- pytorch3d restatement (PARITY UNPINNED: un-vendored dependency, requirements.txt:23) ----------
def euler_xyz_to_matrix(e: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.euler_angles_to_matrix(e, 'XYZ') = Rx(e0) @ Ry(e1) @ Rz(e2)."""
    a, b, c = e.unbind(-1)
    ca, sa, cb, sb, cc, sc = a.cos(), a.sin(), b.cos(), b.sin(), c.cos(), c.sin()
    one, zero = torch.ones_like(a), torch.zeros_like(a)
    rx = torch.stack([one, zero, zero, zero, ca, -sa, zero, sa, ca], -1).reshape(*a.shape, 3, 3)
    ry = torch.stack([cb, zero, sb, zero, one, zero, -sb, zero, cb], -1).reshape(*a.shape, 3, 3)
    rz = torch.stack([cc, -sc, zero, sc, cc, zero, zero, zero, one], -1).reshape(*a.shape, 3, 3)
    return rx @ ry @ rz


def matrix_to_euler_xyz(m: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.matrix_to_euler_angles(m, 'XYZ')."""
    return torch.stack([torch.atan2(-m[..., 1, 2], m[..., 2, 2]), torch.asin(m[..., 0, 2]),
                        torch.atan2(-m[..., 0, 1], m[..., 0, 0])], -1)


def world_to_tcp_frame(action: torch.Tensor, robot_obs: torch.Tensor) -> torch.Tensor:
    """hulc2/models/decoders/utils/gripper_control.py:16-36."""
    b, s, _ = action.shape
    w_T_tcp = euler_xyz_to_matrix(robot_obs[..., 3:6]).float().view(-1, 3, 3)
    tcp_T_w = torch.inverse(w_T_tcp)
    pos = tcp_T_w @ action[..., :3].reshape(-1, 3, 1)
    orn = action[..., 3:6] * 0.01
    w_T_tcp_new = euler_xyz_to_matrix(robot_obs[..., 3:6] + orn).float().view(-1, 3, 3)
    rel = torch.inverse(w_T_tcp_new) @ w_T_tcp
    e = matrix_to_euler_xyz(rel).float()
    e = torch.where(e < -math.pi, e + 2 * math.pi, e)
    e = torch.where(e > math.pi, e - 2 * math.pi, e)
    e = e * 100
    return torch.cat([pos.view(b, s, -1), e.view(b, s, -1), action[..., -1:]], dim=-1)


def tcp_to_world_frame(action: torch.Tensor, robot_obs: torch.Tensor) -> torch.Tensor:
    """gripper_control.py:39-63 (without the NaN quaternion fallback)."""
    b, s, _ = action.shape
    w_T_tcp = euler_xyz_to_matrix(robot_obs[..., 3:6]).float().view(-1, 3, 3)
    pos = w_T_tcp @ action[..., :3].reshape(-1, 3, 1)
    rel = euler_xyz_to_matrix(action[..., 3:6] * 0.01).float().view(-1, 3, 3)
    w_T_new = w_T_tcp @ torch.inverse(rel)
    e = matrix_to_euler_xyz(w_T_new).float() - robot_obs[..., 3:6].reshape(-1, 3)
    e = torch.where(e < -math.pi, e + 2 * math.pi, e)
    e = torch.where(e > math.pi, e - 2 * math.pi, e)
    e = e * 100
    return torch.cat([pos.view(b, s, -1), e.view(b, s, -1), action[..., -1:]], dim=-1)


# ------------------------------------------------------------------------------------------------
# input transforms of the training data pipeline: conf/datamodule/transforms/rand_shift.yaml:1-17
# ------------------------------------------------------------------------------------------------
def random_shifts_aug(x: torch.Tensor, pad: int, shift: torch.Tensor) -> torch.Tensor:
    """RandomShiftsAug.forward, hulc2/utils/transforms.py:85-106, with the torch.randint draw passed in: shift (n, 2) integers
    {sx, sy} in [0, 2*pad].  x (n, c, h, w), h == w."""
    x = x.float()
    n, c, h, w = x.size()
    assert h == w
    x = F.pad(x, (pad,) * 4, "replicate")
    eps = 1.0 / (h + 2 * pad)
    arange = torch.linspace(-1.0 + eps, 1.0 - eps, h + 2 * pad, dtype=x.dtype)[:h]
    arange = arange.unsqueeze(0).repeat(h, 1).unsqueeze(2)
    base_grid = torch.cat([arange, arange.transpose(1, 0)], dim=2).unsqueeze(0).repeat(n, 1, 1, 1)
    sh = shift.reshape(n, 1, 1, 2).to(x.dtype) * (2.0 / (h + 2 * pad))
    return F.grid_sample(x, base_grid + sh, padding_mode="zeros", align_corners=False)


def frames_u8_to_input(frames_u8_nhwc: torch.Tensor, pad: int = 0, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stored uint8 (n, h, w, 3) frames -> the fp32 (n, 3, h, w) tensor in [-1, 1] the model sees: channels first
    (hulc2/datasets/utils/episode_utils.py process_rgb), RandomShiftsAug (train only), ScaleImageTensor (transforms.py:8-19),
    torchvision Normalize(mean 0.5, std 0.5) = (x - 0.5) / 0.5."""
    x = frames_u8_nhwc.permute(0, 3, 1, 2)
    x = random_shifts_aug(x, pad, shift) if shift is not None else x.float()
    x = x.float().div(255)
    return ((x - 0.5) / 0.5).contiguous()


# ------------------------------------------------------------------------------------------------
# play windows: which frames a dataset index yields, and how a short window is padded  
# PARITY UNPINNED for the validation hash only: pyhash (requirements.txt:12, no version) is absent here; FNV-1/32 with the hash
# value starting at the seed (0) is restated from its published algorithm and anchored on pyhash's documented known answer
# fnv1_32()("hello world") == 2805756500 (tests/test_oracle_golden.py).  Everything else is index arithmetic of the reference.
# ------------------------------------------------------------------------------------------------
def fnv1_32(text: str, seed: int = 0) -> int:
    h = seed
    for byte in text.encode("utf-8"):
        h = ((h * 0x01000193) % (1 << 32)) ^ byte
    return h


def get_validation_window_size(idx: int, min_window_size: int, max_window_size: int) -> int:
    """hulc2/datasets/base_dataset.py:26-28"""
    window_range = max_window_size - min_window_size + 1
    return min_window_size + fnv1_32(str(idx)) % window_range


def episode_lookup(ep_start_end_ids, min_window_size: int):
    """hulc2/datasets/utils/shared_memory_loader.py:67-73: (first frame, step inside the episode) of every window start"""
    frames, steps = [], []
    for start_idx, end_idx in ep_start_end_ids:
        for j, idx in enumerate(range(int(start_idx), int(end_idx) + 1 - min_window_size)):
            frames.append(idx)
            steps.append(j)
    return frames, steps


def max_window_size_at(episode_counters, idx: int, min_window_size: int, max_window_size: int) -> int:
    """hulc2/datasets/shm_dataset.py:77-95 — the part of get_window_size before the random / hashed draw"""
    import numpy as np
    episode_counters = np.asarray(episode_counters)
    window_diff = max_window_size - min_window_size
    if len(episode_counters) <= idx + window_diff:
        return min_window_size + len(episode_counters) - idx - 1
    if episode_counters[idx + window_diff] != episode_counters[idx] + window_diff:
        expect = episode_counters[idx] + np.arange(window_diff + 1)
        first_break = np.nonzero(episode_counters[idx : idx + window_diff + 1] - expect)[0][0]
        return min(max_window_size, int(min_window_size + first_break - 1))
    return max_window_size


def minilm_sentence_embedding(sd: SD, input_ids: torch.Tensor, attention_mask: torch.Tensor, p: str = "", num_layers: int = 3, nhead: int = 12,
                              eps: float = 1e-12) -> torch.Tensor:
    """paraphrase-MiniLM-L3-v2 as hulc2/affordance/models/language_encoders/sbert_lang_encoder.py:13-71 runs it: transformers'
    BertModel (embeddings -> 3 x [self-attention, dense + LayerNorm, GELU feed-forward, dense + LayerNorm]) and sentence_transformers'
    mean Pooling.  PARITY: those packages are un-vendored (requirements.txt:19, unpinned); the restatement is pinned against
    transformers' own BertModel with seeded weights (oracle/gen_golden.py minilm -> tests/golden/minilm.npz), not against the
    trained checkpoint, which is not available offline.  sd: BertModel state_dict keys under prefix p; input_ids / attention_mask (B, S)."""
    F = torch.nn.functional
    B, S = input_ids.shape
    x = sd[p + "embeddings.word_embeddings.weight"][input_ids] + sd[p + "embeddings.token_type_embeddings.weight"][0]
    x = x + sd[p + "embeddings.position_embeddings.weight"][:S]
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps)
    hd = D // nhead
    neg = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * torch.finfo(x.dtype).min
    for l in range(num_layers):
        q = p + f"encoder.layer.{l}."
        lin = lambda name, t: F.linear(t, sd[q + name + ".weight"], sd[q + name + ".bias"])
        heads = lambda t: t.view(B, S, nhead, hd).transpose(1, 2)
        qh, kh, vh = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(lin("attention.self.value", x))
        pr = torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5 + neg, dim=-1)
        ctx = (pr @ vh).transpose(1, 2).reshape(B, S, D)
        a = F.layer_norm(lin("attention.output.dense", ctx) + x, (D,), sd[q + "attention.output.LayerNorm.weight"], sd[q + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(lin("intermediate.dense", a))
        x = F.layer_norm(lin("output.dense", h) + a, (D,), sd[q + "output.LayerNorm.weight"], sd[q + "output.LayerNorm.bias"], eps)
    m = attention_mask.to(x.dtype)[:, :, None]
    return (x * m).sum(1) / m.sum(1).clamp(min=1e-9)


def r3m_trunk_features(sd: SD, x: torch.Tensor, p: str = "r3m.convnet.", stages=(2, 2, 2, 2), eps: float = 1e-5) -> torch.Tensor:
    """What `self.r3m(x)` of hulc2/models/perceptual_encoders/vision_r3m.py:24-27 returns for frames x (N,3,H,W) in [0, 255].
    PARITY UNPINNED: `r3m` is an empty, un-vendored submodule (.gitmodules:4-6) and torchvision is absent here; this restates r3m's public
    forward (obs / 255 -> Normalize(ImageNet) -> torchvision resnet18 with fc = Identity, no resize for the default obs_shape) with the
    BatchNorm layers on their running statistics (frozen trunk).  sd: torchvision's ResNet parameter names under prefix p."""
    mean = x.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = x.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)

    def bn(t, q):
        return F.batch_norm(t, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.0, eps)

    t = (x / 255.0 - mean) / std
    t = F.relu(bn(F.conv2d(t, sd[p + "conv1.weight"], None, 2, 3), p + "bn1"))
    t = F.max_pool2d(t, 3, 2, 1)
    for li, n in enumerate(stages, start=1):
        for b in range(n):
            q = p + f"layer{li}.{b}."
            stride = 2 if (b == 0 and li > 1) else 1
            idn = t
            if q + "downsample.0.weight" in sd:
                idn = bn(F.conv2d(t, sd[q + "downsample.0.weight"], None, stride, 0), q + "downsample.1")
            o = F.relu(bn(F.conv2d(t, sd[q + "conv1.weight"], None, stride, 1), q + "bn1"))
            o = bn(F.conv2d(o, sd[q + "conv2.weight"], None, 1, 1), q + "bn2")
            t = F.relu(o + idn)
    return t.mean(dim=(2, 3))


def vision_r3m(sd: SD, p: str, x: torch.Tensor, stages=(2, 2, 2, 2)) -> torch.Tensor:
    """hulc2/models/perceptual_encoders/vision_r3m.py:24-32: frozen trunk (no_grad), flatten, relu(fc1), fc2."""
    with torch.no_grad():
        f = r3m_trunk_features(sd, x, p + "r3m.convnet.", stages)
    return F.linear(F.relu(F.linear(f, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def language_lookup(indx, min_window_size: int, skip_frames: int = 1):
    """hulc2/datasets/npz_dataset.py:182-192 (= shared_memory_loader.py:133-140 for skip_frames 1): window starts of the language
    dataset and the annotation each belongs to; indx = lang_data["info"]["indx"]"""
    episode_lookup, lang_lookup = [], []
    for i, (start_idx, end_idx) in enumerate(indx):
        cnt = 0
        for idx in range(int(start_idx), int(end_idx) + 1 - min_window_size):
            if cnt % skip_frames == 0:
                lang_lookup.append(i)
                episode_lookup.append(idx)
            cnt += 1
    return episode_lookup, lang_lookup


def use_for_aux_lang_loss(lang_lookup, idx: int, aux_lang_loss_window: int) -> bool:
    """hulc2/datasets/shm_dataset.py:150-158"""
    return idx + aux_lang_loss_window < len(lang_lookup) and lang_lookup[idx] < lang_lookup[idx + aux_lang_loss_window]


def pad_with_repetition(x: torch.Tensor, pad_size: int) -> torch.Tensor:
    """hulc2/datasets/base_dataset.py:149-154"""
    return torch.cat([x, x[-1:].expand(pad_size, *x.shape[1:])], dim=0) if pad_size > 0 else x


def pad_with_zeros(x: torch.Tensor, pad_size: int) -> torch.Tensor:
    """hulc2/datasets/base_dataset.py:156-163"""
    return torch.cat([x, torch.zeros(pad_size, x.shape[-1], dtype=x.dtype)], dim=0) if pad_size > 0 else x


def padded_window(frames_u8: Dict[str, torch.Tensor], rel_actions: torch.Tensor, robot_obs: torch.Tensor, start: int, size: int,
                  max_window_size: int) -> Dict[str, torch.Tensor]:
    """hulc2/datasets/base_dataset.py:94-147 for the npz / relative-action dataset (the CALVIN configuration): slice `size` steps
    from `start`, then pad to max_window_size — observations repeat the last step, actions are zero except the repeated gripper dim"""
    pad_size = max_window_size - size
    out = {k: pad_with_repetition(v[start:start + size], pad_size) for k, v in frames_u8.items()}
    a = rel_actions[start:start + size]
    out["actions"] = torch.cat([pad_with_zeros(a[..., :-1], pad_size), pad_with_repetition(a[..., -1:], pad_size)], dim=-1)
    out["robot_obs"] = pad_with_repetition(robot_obs[start:start + size], pad_size)
    return out


# ------------------------------------------------------------------------------------------------
# CLIP-style auxiliary loss
# ------------------------------------------------------------------------------------------------
def clip_auxiliary_loss(sd: SD, seq_feat: torch.Tensor, goal: torch.Tensor, use: Optional[torch.Tensor]) -> torch.Tensor:
    """Hulc2.clip_auxiliary_loss + ProjVisLang.forward, hulc2.py:472-508, proj_vis_lang.py:23-27."""
    if use is not None:
        if not torch.any(use):
            return torch.tensor(0.0)
        seq_feat, goal = seq_feat[use], goal[use]
    im = F.linear(F.relu(F.linear(seq_feat, sd["proj_vis_lang.mlp_im.0.weight"], sd["proj_vis_lang.mlp_im.0.bias"])),
                  sd["proj_vis_lang.mlp_im.2.weight"], sd["proj_vis_lang.mlp_im.2.bias"])
    tx = F.linear(F.relu(F.linear(goal, sd["proj_vis_lang.mlp_lang.0.weight"], sd["proj_vis_lang.mlp_lang.0.bias"])),
                  sd["proj_vis_lang.mlp_lang.2.weight"], sd["proj_vis_lang.mlp_lang.2.bias"])
    im = im / im.norm(dim=-1, keepdim=True)
    tx = tx / tx.norm(dim=-1, keepdim=True)
    logits = sd["logit_scale"].exp() * im @ tx.t()
    labels = torch.arange(logits.shape[0])
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2


# ------------------------------------------------------------------------------------------------
# whole step
# ------------------------------------------------------------------------------------------------
def lmp_train(sd: SD, emb, goal, actions, robot_obs, plan_idx, cfg) -> Dict[str, torch.Tensor]:
    """Hulc2.lmp_train, hulc2.py:200-245, with the categorical sample injected as `plan_idx` (B,32)."""
    pp = plan_proposal(sd, "plan_proposal.", emb[:, 0], goal)
    pr, seq_feat = plan_recognition(sd, "plan_recognition.", emb)
    plan = straight_through_sample(pr, plan_idx)
    lp, ls, mu, grip = decoder_forward(sd, "action_decoder.", plan, emb, goal, emb_slice=cfg.get("emb_slice", (64, 128)))
    acts = world_to_tcp_frame(actions, robot_obs) if cfg.get("gripper_control", False) else actions
    act_loss = decoder_loss(lp, ls, mu, grip, acts)
    kl = kl_loss(pp, pr, cfg.get("kl_beta", 0.01), cfg.get("kl_balancing_mix", 0.8))
    return dict(kl=kl, act=act_loss, total=act_loss + kl, pp=pp, pr=pr, seq_feat=seq_feat, plan=plan)


def real_world_cfg() -> dict:
    """conf/model/real_world_hulc++.yaml:12,20 + conf/model/action_decoder/logistic_decoder_rnn_real_world.yaml:15,19 (BASELINE configs[3])."""
    return dict(gripper_control=False, emb_slice=(0, 128), use_clip_auxiliary_loss=False)


def training_step(sd: SD, batch: Dict[str, Dict], cfg: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Hulc2.training_step, hulc2.py:379-442.  `batch[m]` carries rgb_static, rgb_gripper, actions,
    robot_obs (state_info), plan_idx (injected sample) and, for 'lang', lang (B,384) + use_for_aux_lang_loss."""
    cfg = cfg or {}
    out: Dict[str, torch.Tensor] = {}
    kl = act = total = clip = torch.tensor(0.0)
    for m, db in batch.items():
        emb = concat_encoders(sd, "perceptual_encoder.", db["rgb_static"], db["rgb_gripper"])
        if "lang" in m:
            goal = language_goal_encoder(sd, "language_goal.", db["lang"])
        else:
            goal = visual_goal_encoder(sd, "visual_goal.", emb[:, -1])
        r = lmp_train(sd, emb, goal, db["actions"], db["robot_obs"], db["plan_idx"], cfg)
        if "lang" in m and cfg.get("use_clip_auxiliary_loss", True) and torch.any(db["use_for_aux_lang_loss"]):
            clip = clip + clip_auxiliary_loss(sd, r["seq_feat"], goal, db["use_for_aux_lang_loss"])
        kl, act, total = kl + r["kl"], act + r["act"], total + r["total"]
        for k in ("kl", "act", "total", "pp", "pr", "seq_feat"):
            out[f"{k}_{m}"] = r[k]
        out[f"emb_{m}"], out[f"goal_{m}"] = emb, goal
    n = len(batch)
    total = total / n
    if cfg.get("use_clip_auxiliary_loss", True):
        total = total + cfg.get("clip_auxiliary_loss_beta", 3.0) * clip
    out.update(kl_loss=kl / n, action_loss=act / n, clip_loss=clip, total_loss=total)
    return out
