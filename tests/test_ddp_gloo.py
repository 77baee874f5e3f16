"""CPU multi-process tests (gloo, world_size 2) of the data-parallel gradient path: flat gradient
buffer, bucket construction, hook-driven asynchronous all-reduce in backward-completion order, and
the semantics contract -- k replicas on k micro-batches with per-replica BatchNorm statistics and
averaged gradients.  Gradients are produced by the CPU oracle; the product
code under test is kdrt.optim.FlatParams + kdrt.ddp.BucketedAllReduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import kd_oracle as O
from _util import state_template

FUSION = "weighted"
B, HW, N, G = 1, 32, 96, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _micro_grads(st, seed):
    s = O.clone_state(st, requires_grad=True)
    images, pts, labels = O.make_inputs(B, HW, N, G, seed, pad_tail=8)
    logits, _ = O.complete_model(images, pts, s, fusion_type=FUSION, grid=(G, G), training=True)
    O.weighted_ce(logits, labels, torch.tensor([0.4, 3.5])).backward()
    return s


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    st = O.randomize_state(state_template(FUSION), 5)
    s = O.clone_state(st, requires_grad=True)
    keys = O.trainable_keys(s)
    params = [s[k] for k in keys]
    flat = FlatParams(params)
    red = BucketedAllReduce(flat, keys, n_buckets=3)
    assert len(red.spans) == 3 and red.spans[0][0] == 0 and red.spans[-1][1] == len(keys)
    flat.zero_grad()
    images, pts, labels = O.make_inputs(B, HW, N, G, 100 + rank, pad_tail=8)
    logits, _ = O.complete_model(images, pts, s, fusion_type=FUSION, grid=(G, G), training=True)
    O.weighted_ce(logits, labels, torch.tensor([0.4, 3.5])).backward()
    order = list(red.launch_order)
    scale = red.finish()
    # single-process emulation of the two replicas
    want = torch.zeros_like(flat.grad)
    for r in range(world):
        sr = _micro_grads(st, 100 + r)
        for k, o in zip(keys, flat.offsets):
            want[o:o + sr[k].numel()] += sr[k].grad.reshape(-1)
    err = (flat.grad - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
    # per-replica BN statistics: this rank's running_mean must equal its own micro-batch's, not the other's
    own = _micro_grads(st, 100 + rank)
    bn_key = "camera_encoder.stem.1.running_mean"
    bn_err = (s[bn_key] - own[bn_key]).abs().max().item()
    q.put((rank, err, scale, order, bn_err, all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
                                               for p, o in zip(params, flat.offsets))))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_matches_replica_emulation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale, order, bn_err, views_ok in res:
        assert err < 1e-5, (rank, err)                   # summed gradients == sum over the two replicas
        assert scale == 0.5                              # the optimiser divides by world size
        assert order == [2, 1, 0], order                 # buckets fire head -> fusion/FPN/LiDAR -> camera
        assert bn_err == 0.0                             # BatchNorm statistics stay per replica
        assert views_ok                                  # parameter .grad tensors are views of the flat buffer


def test_flat_params_and_bucket_layout_single_process():
    import sys
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in ((3, 5), (7,), (2, 2, 2), (1,))]
    before = [p.detach().clone() for p in ps]
    flat = FlatParams(ps)
    assert flat.numel % 4 == 0 and all(o % 4 == 0 for o in flat.offsets)       # 16-byte alignment per tensor
    for p, b, o in zip(ps, before, flat.offsets):
        assert torch.equal(p.detach(), b) and p.data_ptr() == flat.data.data_ptr() + 4 * o
    red = BucketedAllReduce(flat, ["a.w", "a.b", "b.w", "c.w"], n_buckets=2)
    assert red.world == 1 and len(red.spans) == 2
    flat.zero_grad()
    sum((p * p).sum() for p in ps).backward()
    assert sorted(red.launch_order) == [0, 1]
    assert red.finish() == 1.0
    for p, o in zip(ps, flat.offsets):
        assert torch.allclose(flat.grad[o:o + p.numel()].view(p.shape), 2 * p.detach())


def test_rank_shard_sampler_equal_steps_for_ragged_frame_counts():
    """Data-parallel training must run the SAME number of steps on every rank (a rank with one batch more would wait
    forever in the bucketed all-reduce): whatever the frame count, every rank gets floor(n / world) training frames,
    disjoint, reshuffled identically per epoch; validation shards cover every frame exactly once (ragged is fine)."""
    from src.data_loading.pandaset_dataset import RankShardSampler
    for n in (7, 8, 9, 13, 101):
        for world in (2, 3, 8):
            for epoch in (0, 1):
                shards = []
                for r in range(world):
                    sm = RankShardSampler(n, r, world, shuffle=True, equal=True, seed=5)
                    sm.set_epoch(epoch)
                    idx = list(sm)
                    assert len(idx) == len(sm) == n // world
                    shards.append(idx)
                flat = [i for sh in shards for i in sh]
                assert len(set(flat)) == len(flat) and set(flat) <= set(range(n))
                for bs in (1, 2, 4):                                   # drop_last batching: equal step counts
                    assert len({len(sh) // bs for sh in shards}) == 1
            e0 = list(RankShardSampler(n, 0, world, True, True, seed=5))
            s1 = RankShardSampler(n, 0, world, True, True, seed=5); s1.set_epoch(1)
            assert n < 2 * world or e0 != list(s1)                     # a new permutation per epoch
            val = [list(RankShardSampler(n, r, world, shuffle=False, equal=False)) for r in range(world)]
            assert sorted(i for v in val for i in v) == list(range(n))
            assert all(len(v) == len(RankShardSampler(n, r, world, False, False)) for r, v in enumerate(val))


def _sink_worker(rank, world, port, q):
    """The product's gradient path without a GPU: backward "kernels" write straight into the flat gradient buffer
    through kdrt.gradsink (out_for / finish / deliver), GradSink.done -> BucketedAllReduce.notify launches each bucket
    when its last gradient has landed.  Ranks hold DIFFERENT numbers of frames (5 vs 3): one step each, sums agree."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
    from kdrt import gradsink
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    names = ["camera_encoder.a", "camera_encoder.b", "fusion.w", "fusion.b", "head.w"]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in ((6, 4), (6,), (5, 6), (5,), (2, 5))]
    flat = FlatParams(params)
    red = BucketedAllReduce(flat, names, n_buckets=3)
    sink = gradsink.install(flat, red)
    frames = 5 if rank == 0 else 3                                     # unequal shards
    x = torch.randn(8, 4, generator=torch.Generator().manual_seed(50))[rank * 5: rank * 5 + frames]

    def grads_of(xr):                                                  # closed-form "backward" of a tiny chain
        return [xr.sum(0).repeat(6, 1) * (i + 1) if p.dim() == 2 and p.shape == (6, 4) else torch.full_like(p, float(i + 1) * xr.shape[0])
                for i, p in enumerate(params)]

    orders = []
    for step in range(2):
        sink.begin_step()
        flat.zero_grad()
        mine = grads_of(x)
        for i in (4, 3, 2, 1, 0):                                      # backward order: head first
            buf, direct = gradsink.out_for(params[i])
            assert direct and buf.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[i]
            if i % 2:
                buf.copy_(mine[i]); assert gradsink.finish(params[i], buf, direct) is None
            else:
                assert gradsink.deliver(params[i], mine[i]) is None
        orders.append(list(red.launch_order))
        scale = red.finish()
        want = torch.zeros_like(flat.grad)
        for r in range(world):
            xr = torch.randn(8, 4, generator=torch.Generator().manual_seed(50))[r * 5: r * 5 + (5 if r == 0 else 3)]
            for gi, o in zip(grads_of(xr), flat.offsets):
                want[o:o + gi.numel()] += gi.reshape(-1)
        err = (flat.grad - want).abs().max().item()
        assert err < 1e-5, err
    # a second gradient for the same parameter inside one step must raise (no silent double counting)
    sink.begin_step()
    b0, d0 = gradsink.out_for(params[0]); gradsink.finish(params[0], b0, d0)
    try:
        gradsink.out_for(params[0])
        twice = False
    except RuntimeError:
        twice = True
    red.finish()
    q.put((rank, orders, scale, twice))
    dist.destroy_process_group()


def test_gradsink_drives_bucketed_allreduce_world2_unequal_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sink_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, orders, scale, twice in res:
        assert orders == [[2, 1, 0], [2, 1, 0]], orders              # head -> fusion -> camera, every step
        assert scale == 0.5 and twice


def test_bucket_fires_only_after_all_its_gradients_with_sink_and_autograd_hooks():
    """The product pattern on CPU: an autograd.Function whose backward writes the parameter gradients into the sink and
    returns None for them.  torch (2.10) still runs the parameters' AccumulateGrad nodes and fires the post-accumulate
    hooks the reducer registered; each parameter must nevertheless count once, so a bucket is launched only when ALL
    its gradients are in the flat buffer (round 1 launched every bucket at the half-way point on the GPU path)."""
    from kdrt import gradsink
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    names = ["enc.w1", "enc.b1", "enc.w2", "enc.b2", "head.w", "head.b"]
    ps = [torch.nn.Parameter(torch.full((4,), float(i + 1))) for i in range(6)]
    flat = FlatParams(ps)
    red = BucketedAllReduce(flat, names, n_buckets=2)
    assert red.spans == [(0, 4), (4, 6)]
    events = []
    launch = red._launch

    def logged(b):
        a, e = red.spans[b]
        events.append(("launch", b, [bool(flat.grad[flat.offsets[i]:flat.offsets[i] + 4].abs().sum() > 0) for i in range(a, e)]))
        launch(b)
    red._launch = logged
    sink = gradsink.install(flat, red)

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.w, ctx.b = w, b
            return x + 1.0

        @staticmethod
        def backward(ctx, g):
            for p in (ctx.w, ctx.b):
                buf, direct = gradsink.out_for(p)
                assert direct
                buf.copy_(torch.full((4,), 3.0))
                events.append(("write", names[red.index_of[id(p)]]))
                assert gradsink.finish(p, buf, direct) is None
            return g, None, None

    try:
        for step in range(2):
            events.clear()
            sink.begin_step()
            flat.zero_grad()
            x = torch.ones(4, requires_grad=True)
            y = Layer.apply(Layer.apply(Layer.apply(x, ps[0], ps[1]), ps[2], ps[3]), ps[4], ps[5])
            y.sum().backward()
            assert red.finish() == 1.0
            launches = [e for e in events if e[0] == "launch"]
            assert [e[1] for e in launches] == [1, 0], events
            assert all(all(e[2]) for e in launches), events            # every gradient of the bucket was already written
            assert events.index(launches[1]) > events.index(("write", "enc.b1"))
            assert torch.equal(flat.grad, torch.full_like(flat.grad, 3.0))
    finally:
        gradsink.uninstall()


def _metrics_worker(rank, world, port, q):
    """Validation with FEWER frames than ranks: rank 1's shard is empty, it never calls update(), and it must still take
    part in the confusion-matrix all-reduce -- otherwise its next collective (here a broadcast) pairs with rank 0's
    all-reduce: a hang or mixed tensors."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src.data_loading.pandaset_dataset import RankShardSampler
    from src.training.trainer import SegmentationMetrics
    n_val = 1                                                        # < world
    mine = list(RankShardSampler(n_val, rank, world, shuffle=False, equal=False))
    m = SegmentationMetrics(num_classes=2, device=torch.device("cpu"))
    for _ in mine:                                                   # what update() leaves behind (its kernel needs a GPU)
        m._dev = torch.tensor([[5, 1], [2, 8]], dtype=torch.int64)
    res = m.compute()
    nxt = torch.tensor([float(rank + 7)])                            # the collective that follows in Trainer.train()
    dist.broadcast(nxt, src=0)
    q.put((rank, len(mine), m.confusion.tolist(), res["miou"], float(nxt)))
    dist.destroy_process_group()


def test_metrics_allreduce_with_an_empty_validation_shard():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [1, 0]                             # rank 1 saw no frame
    for _, _, conf, miou, nxt in res:
        assert conf == [[5, 1], [2, 8]] and nxt == 7.0               # same matrix everywhere; the next collective pairs up
        assert abs(miou - (5 / 8 + 8 / 11) / 2) < 1e-12


def _forced_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = [torch.nn.Parameter(torch.randn(5, generator=torch.Generator().manual_seed(i))) for i in range(4)]
    flat = FlatParams(ps)
    red = BucketedAllReduce(flat, ["a.w", "a.b", "b.w", "b.b"], n_buckets=2, force=True)
    out = []
    for enabled in (True, False, True):
        red.enabled = enabled
        flat.zero_grad()
        sum((p * p).sum() for p in ps).backward()
        issued = red.collectives_issued
        scale = red.finish() if enabled else None
        out.append((enabled, issued, scale, torch.equal(flat.grad[:5], 2 * ps[0].detach())))
    q.put(out)
    dist.destroy_process_group()


def test_forced_reducer_in_a_world_of_one_rank_issues_collectives_and_can_be_silenced():
    """kdrt.ddp.BucketedAllReduce(force=True): the collectives run even with ONE rank (how the RCCL path is executed on
    a 1-GPU box); `enabled = False` silences the hooks for a reducer-less step over the same parameters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(0, 1, _free_port(), q))
    p.start()
    out = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert out == [(True, 2, 1.0, True), (False, 2, None, True), (True, 4, 1.0, True)], out
    from kdrt.ddp import BucketedAllReduce
    from kdrt.optim import FlatParams
    with pytest.raises(RuntimeError):                                # forcing needs a process group
        BucketedAllReduce(FlatParams([torch.nn.Parameter(torch.zeros(4))]), ["w"], force=True)
