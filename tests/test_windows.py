"""Play-window sampling of the HBM-resident episode store: host index logic against the oracle's restatement of
the reference dataset (CPU), device gathers / in-place conv1 reads against the oracle's padded windows (GPU)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from hulc2_amd.datasets import DeviceEpisodeStore, fnv1_32, validation_window_size  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402  (checker only)


def _episodes(rng, n_ep, lo=20, hi=70):
    ids, f = [], 0
    for _ in range(n_ep):
        n = int(rng.integers(lo, hi))
        ids.append((f, f + n - 1))
        f += n
    return ids, f


def _store(n, ids, dev="cpu", hw=(16, 12), validation=False, **kw):
    g = torch.Generator().manual_seed(5)
    rgb = {"rgb_static": torch.randint(0, 256, (n, hw[0], hw[0], 3), generator=g, dtype=torch.uint8),
           "rgb_gripper": torch.randint(0, 256, (n, hw[1], hw[1], 3), generator=g, dtype=torch.uint8)}
    act = torch.rand(n, 7, generator=g) * 2 - 1
    obs = torch.randn(n, 15, generator=g)
    return DeviceEpisodeStore(rgb, act, obs, ids, device=dev, validation=validation, **kw), rgb, act, obs


def test_fnv1_32_known_answer():
    """pyhash's documented example: fnv1_32()('hello world') == 2805756500 (FNV-1, 32 bit, hash value starting at seed 0)"""
    assert fnv1_32(b"hello world") == 2805756500 == O.fnv1_32("hello world")
    assert fnv1_32(b"") == 0
    for idx in (0, 7, 123456):
        assert validation_window_size(idx, 20, 32) == O.get_validation_window_size(idx, 20, 32)
        assert 20 <= validation_window_size(idx, 20, 32) <= 32


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_window_starts_and_sizes_match_reference_logic(seed):
    rng = np.random.default_rng(seed)
    ids, n = _episodes(rng, 6)
    st, *_ = _store(n, ids, validation=True)
    frames, steps = O.episode_lookup(ids, 20)
    assert st.episode_lookup.tolist() == frames and st.episode_counters.tolist() == steps
    for idx in range(len(st)):
        mw = O.max_window_size_at(steps, idx, 20, 32)
        assert st.max_window(idx) == mw
        assert st.get_window_size(idx) == O.get_validation_window_size(idx, 20, mw)
        a = frames[idx]
        ep = next(e for e in ids if e[0] <= a <= e[1])
        assert a + st.get_window_size(idx) - 1 < n
        if len(st) > idx + 12:      # (the reference's "last episode" branch, shm_dataset.py:79-81, ignores boundaries when that episode is short)
            assert a + st.get_window_size(idx) - 1 <= ep[1], "a window never crosses an episode boundary"


def test_training_window_sizes_in_range():
    rng = np.random.default_rng(3)
    ids, n = _episodes(rng, 4)
    st, *_ = _store(n, ids, validation=False, seed=11)
    sizes = [st.get_window_size(i) for i in range(len(st))]
    assert all(20 <= s <= st.max_window(i) for i, s in enumerate(sizes))
    assert len(set(sizes)) > 3
    fixed, *_ = _store(n, ids, min_window_size=32, max_window_size=32)
    assert {fixed.get_window_size(i) for i in range(len(fixed))} == {32}


def _lang_data(rng, ids, n_ann=7, span=64):
    indx = []
    for a, b in ids:
        if b - a + 1 >= span and len(indx) < n_ann:
            s0 = int(rng.integers(a, b + 2 - span))
            indx.append((s0, s0 + span - 1))
    g = torch.Generator().manual_seed(1)
    emb = (torch.randn(len(indx), 1, 384, generator=g) * 0.05).numpy()
    return {"language": {"ann": ["task"] * len(indx), "task": ["t"] * len(indx), "emb": emb}, "info": {"indx": indx}}


def test_language_annotations_follow_the_reference_lookup():
    """auto_lang_ann.npy layout -> window starts, annotation per start, use_for_aux_lang_loss (file-format half)"""
    rng = np.random.default_rng(9)
    ids, n = _episodes(rng, 5, lo=70, hi=120)
    frame0 = 1000                                                      # dataset frame number of store frame 0
    lang = _lang_data(rng, ids)
    lang["info"]["indx"] = [(a + frame0, b + frame0) for a, b in lang["info"]["indx"]]
    _, rgb, act, obs = _store(n, ids)
    st = DeviceEpisodeStore.from_language_annotations(rgb, act, obs, lang, frame0=frame0, device="cpu", validation=True)
    want_lookup, want_lang = O.language_lookup(lang["info"]["indx"], 20)
    assert (st.episode_lookup + frame0).tolist() == want_lookup and st.lang_lookup.tolist() == want_lang
    assert tuple(st.lang_emb.shape) == (len(lang["info"]["indx"]), 384)
    flags = [st.use_for_aux_lang_loss(i) for i in range(len(st))]
    assert flags == [O.use_for_aux_lang_loss(want_lang, i, 8) for i in range(len(st))]
    per_ann = len(want_lang) // len(lang["info"]["indx"])              # 64 - 20 = 44 starts per annotation, the last 8 flagged (not the final one)
    assert sum(flags) == 8 * (len(lang["info"]["indx"]) - 1) and per_ann == 44
    for i in range(len(st)):                                          # windows stay inside their annotated span
        a, b = lang["info"]["indx"][want_lang[i]]
        assert want_lookup[i] + st.get_window_size(i) - 1 <= b or len(st) <= i + 12
    with pytest.raises(NotImplementedError):
        DeviceEpisodeStore.from_language_annotations(rgb, act, obs, lang, frame0=frame0, load_lang_embeddings=False, device="cpu")


def test_bad_window_configuration_raises():
    with pytest.raises(ValueError):
        _store(40, [(0, 39)], min_window_size=33, max_window_size=32)
    with pytest.raises(ValueError):
        _store(40, [(0, 40)])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_device_windows_match_padded_windows(dev):
    """hulc_window_index / hulc_window_rows == the reference's slice + pad_sequence, bit for bit"""
    rng = np.random.default_rng(4)
    ids, n = _episodes(rng, 5)
    st, rgb, act, obs = _store(n, ids, dev=dev, validation=True)
    idxs = rng.integers(0, len(st), 9).tolist() + [len(st) - 1]
    out = st.batch(idxs)
    torch.cuda.synchronize()
    sizes = out["window_sizes"].cpu().tolist()
    assert any(s < 32 for s in sizes)
    for b, (i, size) in enumerate(zip(idxs, sizes)):
        want = O.padded_window(rgb, act, obs, int(st.episode_lookup[i]), size, 32)
        assert torch.equal(out["actions"][b].cpu(), want["actions"])
        assert torch.equal(out["robot_obs"][b].cpu(), want["robot_obs"]) and torch.equal(out["state_info"]["robot_obs"][b].cpu(), want["robot_obs"])
        for k in rgb:
            got = out["rgb_obs"][k][out["rgb_obs"][k + "_index"][b].long()].cpu()
            assert torch.equal(got, want[k])
    assert "rgb_static_shift" not in out["rgb_obs"], "validation transforms have no RandomShiftsAug"


@pytest.mark.gpu
def test_language_batch_on_device(dev):
    """batch() of a language store: the annotation's embedding row and the aux-loss flag of every window start"""
    rng = np.random.default_rng(12)
    ids, n = _episodes(rng, 4, lo=80, hi=130)
    lang = _lang_data(rng, ids)
    _, rgb, act, obs = _store(n, ids)
    st = DeviceEpisodeStore.from_language_annotations(rgb, act, obs, lang, device=dev, validation=True)
    idxs = [0, 35, 36, 43, 44, len(st) - 1]
    out = st.batch(idxs)
    torch.cuda.synchronize()
    emb = torch.as_tensor(lang["language"]["emb"]).reshape(-1, 384)
    assert torch.equal(out["lang"].cpu(), emb[st.lang_lookup[idxs]])
    assert out["use_for_aux_lang_loss"].cpu().tolist() == [O.use_for_aux_lang_loss(st.lang_lookup.tolist(), i, 8) for i in idxs]
    assert out["use_for_aux_lang_loss"].cpu().tolist()[1:3] == [False, True]


@pytest.mark.gpu
@pytest.mark.parametrize("tag,hw,pad", [("static", 200, 10), ("gripper", 84, 4)])
def test_conv1_reads_the_store_in_place(dev, tag, hw, pad):
    """conv1 forward / weight gradient following frame_index == the same kernels on the gathered frames (identical bits)"""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(8)
    store = torch.randint(0, 256, (40, hw, hw, 3), generator=g, dtype=torch.uint8).to(dev)
    index = torch.tensor([3, 4, 5, 5, 5, 39, 0, 17, 17, 2, 38, 39], dtype=torch.int32, device=dev)
    n = index.numel()
    shift = torch.randint(0, 2 * pad + 1, (n, 2), generator=g, dtype=torch.int32).to(dev)
    w = (torch.randn(32, 192, generator=g) / 192 ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    oh = (hw - 8) // 4 + 1
    dy = torch.randn(n, oh, oh, 32, generator=g).to(torch.bfloat16).to(dev)
    res = []
    for x, ix in ((store[index.long()].contiguous(), None), (store, index)):
        y = torch.empty(n, oh, oh, 32, device=dev, dtype=torch.bfloat16)
        kn.conv2d_fwd(x, w, b, y, n, hw, hw, 3, 32, 8, 8, 4, True, relu=True, aug_shift=shift, aug_pad=pad, frame_index=ix)
        dw, db = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
        kn.conv2d_bwd_weight(x, dy, dw, db, n, hw, hw, 3, 32, 8, 8, 4, True, aug_shift=shift, aug_pad=pad, frame_index=ix)
        res.append((y, dw, db))
    torch.cuda.synchronize()
    for a, c in zip(*res):
        assert torch.equal(a, c)
    with pytest.raises(TypeError):
        kn.conv2d_fwd(store.float().permute(0, 3, 1, 2).contiguous(), w, b, res[0][0], n, hw, hw, 3, 32, 8, 8, 4, True, frame_index=index)


@pytest.mark.gpu
def test_training_step_from_the_store(dev, monkeypatch):
    """Hulc2.training_step on store-backed windows (index rows, nothing gathered) == on the materialised padded uint8 windows, bit for bit.
    (conv1 launched per modality on both sides: two materialised tensors would otherwise share ONE launch — another summation order of the
    weight gradient's slabs — while windows of two different stores cannot; the one-launch forms have their own test in test_conv_gpu.py)"""
    monkeypatch.setenv("HULC_CONV1_PER_INPUT", "1")
    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config

    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 9)
    m.train()
    rng = np.random.default_rng(6)
    ids, n = _episodes(rng, 3, lo=30, hi=50)
    g = torch.Generator().manual_seed(2)
    lang_emb = torch.randn(5, 384, generator=g) * 0.05
    B, S = 2, 32
    stores = {}
    for mod in ("vis", "lang"):
        st, *_ = _store(n, ids, dev=dev, hw=(200, 84), seed=3)
        if mod == "lang":
            st = DeviceEpisodeStore(st.rgb, st.rel_actions, st.robot_obs, ids, device=dev, lang_emb=lang_emb,
                                    lang_lookup=rng.integers(0, 5, len(st)).tolist(), seed=4)
        stores[mod] = st
    idxs = {"vis": [1, len(stores["vis"]) - 2], "lang": [5, 9]}
    by_index, gathered = {}, {}
    for mod, st in stores.items():
        db = st.batch(idxs[mod])
        db = {k: (dict(v) if isinstance(v, dict) else v) for k, v in db.items()}
        by_index[mod] = db
        gd = dict(db)
        gd["rgb_obs"] = {}
        for k in st.rgb:
            ix = db["rgb_obs"][k + "_index"].long()
            gd["rgb_obs"][k] = st.rgb[k][ix].contiguous()                       # (B, S, H, W, 3) uint8, padded by repetition
            gd["rgb_obs"][k + "_shift"] = db["rgb_obs"][k + "_shift"]
        gathered[mod] = gd
    assert (by_index["vis"]["window_sizes"] < S).any()
    outs = []
    for bt in (gathered, by_index):
        for p in m.parameters():
            p.grad = None
        kn.reset_step_state(dev) if hasattr(kn, "reset_step_state") else None
        loss = m.training_step(bt, 0)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) == len(outs[1][1]) > 0
    for a, c in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, c)
