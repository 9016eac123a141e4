"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py / uformer_amd.dist.
Inference shards the batch with NO data-path collective; ranks only meet for the wall-clock
(max over ranks) and for an optional output gather.  The per-rank compute is stood in for by the
oracle (tests may use it), so the sharded result must equal the single-process result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import uformer_oracle as O
    from uformer_amd import dist as ud
    from uformer_amd import spec
    torch.set_num_threads(2)
    r, lr, w = ud.init_process_group("gloo")
    assert (r, w) == (rank, world)
    cfg = spec.arch_config("tiny", 128)
    sd = spec.synth_state_dict(cfg, 5)
    gb = 3                                             # uneven: rank 0 gets 2 images, rank 1 gets 1
    x = spec.synth_input(gb, 128, 128, 6)
    a, b = ud.shard_batch(gb, rank, world)
    y = O.uformer_forward(x[a:b], sd, img_size=128, embed_dim=16, depths=cfg.depths, num_heads=cfg.num_heads)
    ud.barrier()
    t = ud.max_over_ranks(1.0 + rank)                  # slowest rank defines the step time
    n = ud.sum_over_ranks(float(b - a))
    full = ud.gather_batch(y, gb)
    if rank == 0:
        ref = O.uformer_forward(x, sd, img_size=128, embed_dim=16, depths=cfg.depths, num_heads=cfg.num_heads)
        q.put((t, n, float((full - ref).abs().max())))
    dist.destroy_process_group()


def test_two_rank_sharded_inference_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, n, err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0 and n == 3.0
    assert err < 1e-5


def _train_worker(rank, world, port, q):
    """Data-parallel training exchange: each rank differentiates the loss of ITS shard (oracle autograd stands in for the HIP
    backward), GradientAllReduce averages; with the per-rank loss weighted by its shard size the result must equal the
    single-process gradient of the whole batch."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import uformer_oracle as O
    from uformer_amd import dist as ud
    from uformer_amd import spec
    torch.set_num_threads(2)
    ud.init_process_group("gloo")
    cfg = spec.arch_config("tiny", 128)
    kw = dict(img_size=128, embed_dim=16, depths=cfg.depths, num_heads=cfg.num_heads)
    gb = 2
    x, tgt = spec.synth_input(gb, 128, 128, 6), spec.synth_input(gb, 128, 128, 7)

    def grads_of(xs, ts, weight):
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 5).items()}
        (O.charbonnier_loss(O.uformer_forward(xs, sd, **kw), ts) * weight).backward()
        return [v for v in sd.values() if v.is_floating_point()]

    a, b = ud.shard_batch(gb, rank, world)
    params = grads_of(x[a:b], tgt[a:b], world * (b - a) / gb)       # mean over ranks of (world * n_r / N) * L_r == L of the whole batch
    if rank == 1:
        params[3].grad = None                                          # a gradient DropPath removed on one rank only
    red = ud.GradientAllReduce(params, bucket_bytes=256 << 10)
    assert len(red.buckets) > 3 and sum(len(bk) for bk in red.buckets) == len(params)
    red()
    if rank == 0:
        ref = grads_of(x, tgt, 1.0)
        keep = grads_of(x[a:b], tgt[a:b], world * (b - a) / gb)[3].grad / world      # rank 1 contributed zeros to this one
        err = max(float((p.grad - r.grad).abs().max() / r.grad.abs().max()) for i, (p, r) in enumerate(zip(params, ref)) if i != 3)
        err3 = float((params[3].grad - keep).abs().max() / keep.abs().max())
        q.put((err, err3, len(red.buckets)))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, err3, nb = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-4 and err3 < 1e-5 and nb > 3


def _overlap_worker(rank, world, port, q, gb=2, dropped_by_rank=None, algorithm="ring", payload="float32"):
    """OverlappedGradientAllReduce driven the way UformerTape.backward drives it: gradients arrive stage by stage in reverse-sweep
    order (head, decoder 3..0, bottleneck, encoder 3..0, stem), every bucket is reduced the moment its last gradient is in, p.grad
    is a view of the bucket, and the average is folded into the optimizer's grad_scale."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import uformer_oracle as O
    from uformer_amd import dist as ud
    from uformer_amd import spec
    torch.set_num_threads(2)
    ud.init_process_group("gloo")
    cfg = spec.arch_config("tiny", 128)
    kw = dict(img_size=128, embed_dim=16, depths=cfg.depths, num_heads=cfg.num_heads)
    x, tgt = spec.synth_input(gb, 128, 128, 6), spec.synth_input(gb, 128, 128, 7)

    def grads_of(xs, ts, weight):
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in spec.synth_state_dict(cfg, 5).items()}
        (O.charbonnier_loss(O.uformer_forward(xs, sd, **kw), ts) * weight).backward()
        return {k: v for k, v in sd.items() if v.is_floating_point()}

    a, b = ud.shard_batch(gb, rank, world)
    mine = grads_of(x[a:b], tgt[a:b], world * (b - a) / gb)
    params = {k: torch.nn.Parameter(v.detach().clone()) for k, v in mine.items()}
    sink = ud.OverlappedGradientAllReduce(list(params.items()), bucket_bytes=256 << 10, algorithm=algorithm, payload=getattr(torch, payload))
    assert all(p.grad is not None and p.grad.data_ptr() == sink.views[k].data_ptr() for k, p in params.items())
    stages = ["output_proj", "decoderlayer_3", "upsample_3", "decoderlayer_2", "upsample_2", "decoderlayer_1", "upsample_1", "decoderlayer_0", "upsample_0",
              "conv", "dowsample_3", "encoderlayer_3", "dowsample_2", "encoderlayer_2", "dowsample_1", "encoderlayer_1", "dowsample_0", "encoderlayer_0",
              "input_proj"]
    # gradients DropPath removed locally: different sets on different ranks (uneven masks) -- the collectives must still be identical
    dropped_by_rank = dropped_by_rank or {1: ["encoderlayer_2.blocks.0.mlp.linear1.0.weight"]}
    dropped = set(dropped_by_rank.get(rank, []))
    # what the exchange must produce, by plain (unbucketed) all-reduces of the same local gradients: the sink's reference
    want = {}
    for k, v in mine.items():
        w = torch.zeros_like(v.grad) if k in dropped else v.grad.clone()
        dist.all_reduce(w)
        want[k] = w / world
    sink.begin_step()
    launched_at = []
    for st in stages:
        group = {k: (None if k in dropped else v.grad) for k, v in mine.items() if k.split(".")[0] == st}
        assert group, st
        sink.deliver(group)
        launched_at.append(len(sink.launch_order))
    sink.finish()
    assert sorted(sink.launch_order) == list(range(len(sink.buckets)))
    assert launched_at[len(stages) // 2] >= 1 and launched_at[len(stages) // 2] < len(sink.buckets)     # collectives start mid-sweep
    rows = ud.rank_devices("cpu")                                        # what bench.py puts into config.rank_devices: one row per rank that met
    assert [r[0] for r in rows] == list(range(world)) and all(len(r) == 4 for r in rows)
    orders = [None] * world
    dist.all_gather_object(orders, list(sink.launch_order))
    assert all(o == orders[0] for o in orders), orders                   # every rank issued the same collectives in the same order
    err = 0.0
    for k, p in params.items():
        got = p.grad * sink.grad_scale
        err = max(err, float((got - want[k]).abs().max() / max(float(want[k].abs().max()), 1e-30)))
    # replicas must hold the SAME bits after the exchange (a 2-byte wire type rounds the reduced slice once, for its owner too)
    digest = torch.cat([p.grad.reshape(-1) for p in params.values()]).double()
    sums = [None] * world
    dist.all_gather_object(sums, (float(digest.sum()), float((digest * digest).sum())))
    assert all(s_ == sums[0] for s_ in sums), sums
    if rank == 0:
        nodrop = all(not v for v in dropped_by_rank.values())
        if nodrop or world == 2:      # also against the single-process gradient of the whole batch where no rank dropped that tensor
            ref = grads_of(x, tgt, 1.0)
            alld = set(sum(dropped_by_rank.values(), []))
            for k, p in params.items():
                if k not in alld:
                    err = max(err, float((p.grad * sink.grad_scale - ref[k].grad).abs().max() / ref[k].grad.abs().max()))
        q.put((err, len(sink.buckets), list(sink.launch_order)))
    dist.destroy_process_group()


def test_two_rank_overlapped_allreduce_delivers_in_sweep_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, nb, order = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-4 and nb > 3
    assert order[0] == 0        # the bucket holding the decoder's last blocks (first to finish in the reverse sweep) goes first


def test_three_rank_overlapped_allreduce_uneven_droppath_masks():
    """Three ranks, one image each, a different set of locally dropped gradients on every rank (uneven DropPath masks): each rank
    launches the same buckets in the same order (checked with all_gather_object inside the workers) and the exchanged gradients
    equal plain per-tensor all-reduces of the same local gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    dropped = {0: [], 1: ["encoderlayer_2.blocks.0.mlp.linear1.0.weight", "decoderlayer_0.blocks.0.attn.proj.weight"],
               2: ["decoderlayer_3.blocks.0.modulator.weight", "encoderlayer_2.blocks.0.mlp.linear1.0.weight", "conv.blocks.0.norm1.bias"]}
    procs = [ctx.Process(target=_overlap_worker, args=(r, 3, port, q, 3, dropped)) for r in range(3)]
    for p in procs:
        p.start()
    err, nb, order = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5 and nb > 3 and order[0] == 0


def test_two_rank_direct_exchange_equals_the_allreduce():
    """algorithm="direct" (all_to_all_single + f32 sum on receipt + all_gather_into_tensor) with an f32 wire type: the same gradients as the
    bucketed all-reduce, the same launch order, replicas bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q, 2, None, "direct", "float32")) for r in range(2)]
    for p in procs:
        p.start()
    err, nb, order = q.get(timeout=400)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-4 and nb > 3 and order[0] == 0


def test_three_rank_direct_exchange_bf16_wire_f32_accumulation():
    """Three ranks (a world that does not divide the bucket: padded slices), bf16 on the wire, uneven DropPath masks: within bf16 rounding of the f32
    all-reduce (two roundings: the sender's slice and the reduced slice), and every rank ends with identical bits."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    dropped = {0: [], 1: ["encoderlayer_2.blocks.0.mlp.linear1.0.weight"], 2: ["conv.blocks.0.norm1.bias"]}
    procs = [ctx.Process(target=_overlap_worker, args=(r, 3, port, q, 3, dropped, "direct", "bfloat16")) for r in range(3)]
    for p in procs:
        p.start()
    err, nb, order = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1.2e-2 and nb > 3 and order[0] == 0


def _traj_worker(rank, world, port, q, algorithm):
    """K optimizer steps of data-parallel training through the gradient sink: every rank differentiates the loss of ITS half of the batch (oracle autograd
    stands in for the HIP backward), the sink exchanges the bucket views that ARE param.grad, and the 1 / world of the average is folded into the update as
    uf_adamw_step's grad_scale does (here: one scaling of the bucket before torch.optim.AdamW).  Rank 0 also runs the same K steps alone on the whole batch."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import uformer_oracle as O
    from uformer_amd import dist as ud
    from uformer_amd import spec
    torch.set_num_threads(2)
    ud.init_process_group("gloo")
    cfg = spec.arch_config("tiny", 128)
    kw = dict(img_size=128, embed_dim=16, depths=cfg.depths, num_heads=cfg.num_heads)
    K, gb = 3, 2
    xs = [spec.synth_input(gb, 128, 128, 60 + k) for k in range(2)]          # two alternating batches, as the reference-trajectory fixture
    ts = [spec.synth_input(gb, 128, 128, 70 + k) for k in range(2)]

    def run(shard, use_sink):
        sd0 = spec.synth_state_dict(cfg, 5)
        params = {k: torch.nn.Parameter(v.clone()) for k, v in sd0.items() if v.is_floating_point()}
        full = dict(sd0, **params)
        opt = torch.optim.AdamW(list(params.values()), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)      # train/train_denoise.py:77
        sink = ud.OverlappedGradientAllReduce(list(params.items()), bucket_bytes=256 << 10, algorithm=algorithm) if use_sink else None
        losses = []
        for k in range(K):
            a, b = shard
            if sink is not None:
                sink.begin_step()
            else:
                opt.zero_grad(set_to_none=True)
            loss = O.charbonnier_loss(O.uformer_forward(xs[k % 2][a:b], full, **kw), ts[k % 2][a:b])
            grads = torch.autograd.grad(loss, list(params.values()))
            if sink is not None:
                sink.deliver({n: g_ for n, g_ in zip(params.keys(), grads)})
                sink.finish()
                for p in params.values():
                    assert p.grad.data_ptr() == sink.views[[n for n, q_ in params.items() if q_ is p][0]].data_ptr()
                for f in sink.flat:
                    f.mul_(sink.grad_scale)                              # uf_adamw_step(grad_scale = 1 / world) reads the bucket and scales in its registers
            else:
                for p, g_ in zip(params.values(), grads):
                    p.grad = g_
            opt.step()
            losses.append(float(loss))
        return {n: p.detach().clone() for n, p in params.items()}, losses

    a, b = ud.shard_batch(gb, rank, world)
    mine, my_losses = run((a, b), True)
    # replicas hold identical weights after K steps
    digest = float(torch.cat([v.reshape(-1) for v in mine.values()]).double().sum())
    all_d = [None] * world
    dist.all_gather_object(all_d, digest)
    assert all(d == all_d[0] for d in all_d), all_d
    if rank == 0:
        ref, ref_losses = run((0, gb), False)
        err = max(float((mine[n] - ref[n]).abs().max() / max(float(ref[n].abs().max()), 1e-12)) for n in ref)
        moved = max(float((ref[n] - spec.synth_state_dict(cfg, 5)[n]).abs().max()) for n in ref)
        q.put((err, moved, my_losses, ref_losses))
    dist.destroy_process_group()


def test_two_rank_training_trajectory_equals_single_process():
    """bucket views -> exchange -> update with grad_scale = 1 / world over K = 3 steps on 2 ranks x half the batch gives the weights of
    1 rank x the whole batch (mean loss: the average of the two half-batch gradients IS the full-batch gradient), with the ring all-reduce and with the
    direct all-to-all exchange."""
    for algorithm in ("ring", "direct"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_traj_worker, args=(r, 2, port, q, algorithm)) for r in range(2)]
        for p in procs:
            p.start()
        err, moved, my_losses, ref_losses = q.get(timeout=900)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert moved > 1e-4                       # the weights did move (3 AdamW steps at lr 2e-4)
        assert err < 2e-5, (algorithm, err)
        assert len(my_losses) == 3 and len(ref_losses) == 3
