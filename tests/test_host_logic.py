"""CPU: host-side logic of the boundary -- state_dict layout, constructor semantics, weight
re-layouts (checked against the oracle's arithmetic on CPU tensors), sharding helpers."""
import numpy as np
import torch

from oracle import uformer_oracle as O
from uformer_amd import dist as ud
from uformer_amd import model, packing, spec


def test_state_dict_layout_all_archs():
    for arch, n in (("Uformer_T", None), ("Uformer_S", None), ("Uformer_B", 759)):
        m = model.get_arch(arch, 128)
        cfg = m.cfg
        sp = spec.state_dict_spec(cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _, _ in sp]
        for k, shape, kind in sp:
            assert tuple(sd[k].shape) == tuple(shape), k
            assert sd[k].dtype == (torch.int64 if kind == "rpi" else torch.float32), k
        if n:
            assert len(sd) == n
    assert sum(p.numel() for p in model.get_arch("Uformer_B", 128).parameters()) == 50880946


def test_ctor_clamp_and_modulator_placement():
    m = model.get_arch("Uformer_B", 128)
    assert [b.shift_size for b in m.conv.blocks] == [0, 0]            # model.py:863-866 at 8x8
    m = model.get_arch("Uformer_B", 256)
    assert [b.shift_size for b in m.conv.blocks] == [0, 4]
    assert [b.shift_size for b in m.encoderlayer_2.blocks] == [0, 4] * 4
    assert m.encoderlayer_0.blocks[0].modulator is None                # encoder: no modulator
    assert m.decoderlayer_3.blocks[0].modulator.weight.shape == (64, 64)
    assert m.cfg.block_shifts() == [[b.shift_size for b in getattr(m, s).blocks] for s in spec.STAGES]
    assert abs(m.flops() / 1e9 - 86.574) < 1e-2                        # exact MACs, SURVEY section 8d


def test_load_checkpoint_forms():
    cfg = spec.arch_config("tiny", 128)
    sd = spec.synth_state_dict(cfg, 3)
    m = model.Uformer(img_size=128, embed_dim=16, depths=[1] * 9, modulator=True)
    m.load_state_dict(sd, strict=True)
    m.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    m.load_state_dict({"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": {}}, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])


def test_weight_packing_matches_reference_arithmetic():
    g = torch.Generator().manual_seed(0)
    # downsample: Conv2d(k4,s2,p1) == im2col GEMM with k = (ky*4+kx)*C + c
    w = torch.randn(16, 8, 4, 4, generator=g)
    x = torch.randn(1, 8, 8, 8, generator=g)                    # NCHW
    ref = torch.nn.functional.conv2d(x, w, stride=2, padding=1)
    cols = torch.nn.functional.unfold(x, 4, padding=1, stride=2)  # (1, C*16, L) with (c,ky,kx) order
    cols = cols.reshape(1, 8, 16, -1).permute(0, 3, 2, 1).reshape(-1, 16 * 8)  # -> (ky,kx,c)
    got = (cols @ packing.pack_downsample(w, torch.float32).t()).t().reshape(1, 16, 4, 4)
    assert (got - ref).abs().max() < 1e-4
    # upsample: ConvTranspose2d(k2,s2) == GEMM with n = (dy*2+dx)*Cout + co then scatter
    wu = torch.randn(8, 4, 2, 2, generator=g)
    xu = torch.randn(1, 8, 3, 3, generator=g)
    refu = torch.nn.functional.conv_transpose2d(xu, wu, stride=2)
    y = xu.permute(0, 2, 3, 1).reshape(-1, 8) @ packing.pack_upsample(wu, torch.float32).t()   # (9, 4*4)
    gotu = y.reshape(3, 3, 2, 2, 4).permute(4, 0, 2, 1, 3).reshape(1, 4, 6, 6)
    assert (gotu - refu).abs().max() < 1e-5
    # dense relative-position bias == the reference gather
    table = torch.randn(225, 3, generator=g)
    idx = spec.relative_position_index(8)
    assert torch.equal(packing.rpb_dense(table, idx), O.relative_position_bias(table, idx))
    # depthwise taps, stem and head layouts
    wd = torch.randn(6, 1, 3, 3, generator=g)
    assert torch.equal(packing.pack_dwconv(wd)[4], wd[:, 0, 1, 1])
    wi = torch.randn(8, 3, 3, 3, generator=g)
    assert torch.equal(packing.pack_input_proj(wi)[1 * 9 + 2 * 3 + 0], wi[:, 1, 2, 0])
    wo = torch.randn(3, 16, 3, 3, generator=g)
    assert torch.equal(packing.pack_output_proj(wo)[2, 1 * 3 + 2], wo[2, :, 1, 2])


def test_window_index_closed_form_is_a_permutation():
    for B, H, W, s in ((2, 16, 24, 4), (1, 8, 8, 0), (3, 32, 16, 4)):
        idx = O.window_partition_index(B, H, W, 8, s)
        assert np.array_equal(np.sort(idx), np.arange(B * H * W))


def test_shard_batch():
    for gb, w in ((16, 1), (16, 8), (17, 4), (3, 8), (256, 8)):
        parts = [ud.shard_batch(gb, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == gb
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1


def test_infer_wrapper_has_no_cpu_path_and_tiles_cover_the_frame():
    """uformer_amd.infer pads / crops with kernels (tests/test_gpu_tail.py checks them against the reference helper restated in
    the oracle); on CPU tensors it raises instead of falling back.  The tile placement of restore_tiled is host logic: tiles
    cover the frame, start and end on it, and overlap by at least the requested margin."""
    import pytest
    import torch
    from uformer_amd import infer
    from uformer_amd._lib import UformerHipError
    with pytest.raises(UformerHipError):
        infer.expand2square(torch.rand(1, 3, 40, 24), 16.0)
    for (n, tile, ov) in ((1280, 768, 128), (720, 768, 128), (2000, 512, 128), (769, 768, 256), (1536, 768, 0)):
        st = infer._starts(n, tile, ov)
        assert st[0] == 0 and (n <= tile or st[-1] + tile == n)
        for a, b in zip(st, st[1:]):
            assert 0 < b - a <= tile - ov, (n, tile, ov, st)


def test_fragment_major_packing_formula_cpu():
    """packing.pack_frag == the layout formula printed in include/uformer_hip.h, element by element (also with K padded
    up to a multiple of 32), and packing.pack_rpb_table round-trips a Toeplitz bias and rejects a non-Toeplitz one."""
    import torch
    from uformer_amd import packing, spec
    for (N, K) in ((32, 64), (48, 16), (16, 96)):
        w = torch.arange(N * K, dtype=torch.float32).reshape(N, K)
        fm = packing.pack_frag(w, torch.float32).reshape(-1)
        KS = (K + 31) // 32
        assert fm.numel() == (N // 16) * KS * 64 * 8
        for n in range(0, N, 5):
            for k in range(0, KS * 32, 3):
                idx = ((n // 16 * KS + k // 32) * 64 + ((k % 32) // 8) * 16 + n % 16) * 8 + k % 8
                assert fm[idx].item() == (w[n, k].item() if k < K else 0.0), (N, K, n, k)
    heads = 2
    table = torch.randn(225, heads)
    idx = spec.relative_position_index(8)
    dense = packing.rpb_dense(table, idx)
    tab = packing.pack_rpb_table(dense)
    assert tab is not None and tab.shape == (heads, 15, 15)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    for q in (0, 9, 37, 63):
        for k in (0, 7, 28, 63):
            assert tab[1, ys[q] - ys[k] + 7, 7 - (xs[q] - xs[k])] == dense[1, q, k]
    broken = dense.clone(); broken[0, 3, 5] += 1.0
    assert packing.pack_rpb_table(broken) is None


def test_block_gradient_buffer_covers_every_parameter_of_every_block():
    """ops._block_grads lays the uf_block_grads outputs out in one flat buffer; train._named_block_grads maps them onto the
    reference's parameter names: for every LeWin block of Uformer-B the names and shapes must be exactly the block's float
    parameters (the fused q|k|v gradient split into to_q / to_kv), every piece 256-byte aligned."""
    import torch
    from uformer_amd import ops, spec, train
    cfg = spec.arch_config("Uformer_B", 128)
    shapes = {k: tuple(shape) for k, shape, kind in spec.state_dict_spec(cfg) if kind != "rpi"}
    dims = [cfg.embed_dim * m for m in (1, 2, 4, 8, 16, 16, 8, 4, 2)]
    seen = set()
    for s, stage in enumerate(spec.STAGES):
        C, heads = dims[s], cfg.num_heads[s]
        for b in range(cfg.depths[s]):
            prefix = f"{stage}.blocks.{b}."
            has_mod = (prefix + "modulator.weight") in shapes
            flat, g, views = ops._block_grads(C, heads, has_mod, "cpu")
            assert all(v.data_ptr() % 256 == flat.data_ptr() % 256 for v in views.values())
            assert (g.modulator is not None) == has_mod
            named = train._named_block_grads(prefix, views, C)
            want = {k for k in shapes if k.startswith(prefix)}
            assert set(named) == want, (prefix, set(named) ^ want)
            for k, v in named.items():
                assert tuple(v.shape) == shapes[k], (k, tuple(v.shape), shapes[k])
            seen |= want
    rest = {k.split(".")[0] for k in shapes if k not in seen}
    assert rest == {"input_proj", "output_proj", "dowsample_0", "dowsample_1", "dowsample_2", "dowsample_3", "upsample_0", "upsample_1", "upsample_2", "upsample_3"}


def test_relative_position_index_check_accepts_only_the_reference_buffer():
    import torch
    from uformer_amd import spec, train
    idx = spec.relative_position_index(8)
    assert train._index_is_standard(idx.clone())
    bad = idx.clone(); bad[3, 5] += 1
    assert not train._index_is_standard(bad)


def test_gradient_sink_step_boundaries_and_accumulation():
    """OverlappedGradientAllReduce state machine: a loop that forgets begin_step() still gets THIS step's gradients
    (the first deliver after finish starts a new step); begin_step(accumulate=2) adds two backward passes and launches the
    collectives during the second; one pass too many raises instead of being dropped or double counted."""
    import pytest
    import torch
    from uformer_amd import dist as ud
    ps = [("a", torch.nn.Parameter(torch.zeros(300))), ("b", torch.nn.Parameter(torch.zeros(500))), ("c", torch.nn.Parameter(torch.zeros(7)))]
    sink = ud.OverlappedGradientAllReduce(ps, bucket_bytes=2048)
    assert len(sink.buckets) >= 2
    g1 = {n: torch.full_like(p, 1.0 + i) for i, (n, p) in enumerate(ps)}
    g2 = {n: torch.full_like(p, 10.0 + i) for i, (n, p) in enumerate(ps)}
    sink.deliver(g1); sink.finish()
    assert all(torch.equal(p.grad, g1[n]) for n, p in ps)
    sink.deliver(g2); sink.finish()                                   # no begin_step(): must not keep step 1's gradients
    assert all(torch.equal(p.grad, g2[n]) for n, p in ps)
    assert sorted(sink.launch_order) == list(range(len(sink.buckets)))
    sink.deliver({"a": g1["a"]}); sink.finish()                       # parameters without a gradient this step are zero, not stale
    assert torch.equal(ps[0][1].grad, g1["a"]) and float(ps[1][1].grad.abs().sum()) == 0.0
    sink.begin_step(accumulate=2)
    sink.deliver(g1)
    assert sink.launch_order == []                                    # nothing is reduced before the last pass
    sink.deliver({"a": g2["a"], "b": None, "c": g2["c"]})              # None in a later pass adds nothing
    assert sorted(sink.launch_order) == list(range(len(sink.buckets)))
    sink.finish()
    assert torch.equal(ps[0][1].grad, g1["a"] + g2["a"]) and torch.equal(ps[1][1].grad, g1["b"]) and torch.equal(ps[2][1].grad, g1["c"] + g2["c"])
    sink.begin_step()
    sink.deliver(g1)
    with pytest.raises(RuntimeError, match="accumulate"):
        sink.deliver(g2)


def test_restore_tiled_seams_are_exact_for_a_pointwise_model():
    """restore_tiled's cutting, zero padding, ramp weights and normalisation (VERDICT r02 "weak" 9): with a model that acts pixel by
    pixel, tiling must reproduce the full-frame result EXACTLY (to blending round-off) at every pixel, seams and borders
    included -- a misplaced tile, a wrong ramp or an unnormalised overlap shows up at O(0.1).  (With the real network the tiled
    result is an approximation by construction: its receptive field spans several hundred pixels.)  Pure torch: runs on the CPU."""
    import torch
    from uformer_amd import infer
    calls = []

    def pointwise(t):
        calls.append(tuple(t.shape))
        return 0.5 * t + 0.125 * t * t + 0.1

    g = torch.Generator().manual_seed(3)
    for (h, w, tile, ov, mb) in ((384, 640, 256, 64, 4), (300, 517, 128, 32, 3), (130, 900, 128, 64, 16)):
        img = torch.rand(2, 3, h, w, generator=g)
        calls.clear()
        got = infer.restore_tiled(pointwise, img, tile=tile, min_overlap=ov, max_batch=mb, clamp=False)
        assert got.shape == img.shape and all(c[-1] == tile and c[-2] == tile for c in calls)
        assert (got - pointwise(img)).abs().max().item() < 2e-6, (h, w, tile)
        assert len(calls) > 1
