"""tf_comm_* / tf_allreduce_bucket on the device: a one-rank RCCL communicator behind the C ABI (the one GPU these tests get; RCCL
refuses two ranks on one device, so N > 1 through this entry is covered by construction and by the gloo tests of the code around it).
One rank still exercises everything on this side of the wire: the run-time binding of librccl, communicator creation on the current
device, the in-place all-reduce enqueued on a stream that is NOT the current one, ordering against work on that stream, the counters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_allreduce_is_in_place_identity_and_stream_ordered():
    from transfusion_amd.comm import BucketComm
    dev = torch.device("cuda", 0)
    comm = BucketComm(1, 0, BucketComm.new_unique_id(), dev)
    try:
        side = torch.cuda.Stream(device=dev)
        g = torch.zeros(3_000_000, device=dev)
        want = torch.randn(3_000_000, generator=torch.Generator().manual_seed(5)).to(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            g.copy_(want)                                    # producer on the side stream ...
        comm.all_reduce_(g[1024:2_000_000], stream=side)     # ... the collective behind it on the same stream
        comm.all_reduce_(g[:1024], stream=side)
        torch.cuda.current_stream(dev).wait_stream(side)
        assert torch.equal(g, want)                          # sum over one rank
        st = comm.stats()
        assert st == {"world": 1, "rank": 0, "calls": 2, "elems": 2_000_000}
        with pytest.raises(Exception):
            comm.all_reduce_(g.to(torch.bfloat16))           # fp32 only
    finally:
        comm.close()


def test_train_step_accepts_comm_choice_at_world_one():
    from cases import make_encoder_inputs  # noqa: F401  (path set up by conftest)
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    from transfusion_amd.runner.trainer import FusionTrainStep
    dev = torch.device("cuda", 0)
    enc = CrossTransformerModuleBox(no_patches=64, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 64, 64), num_layers=1,
                                    patch_dropout=0.0, num_heads=2, token_dropout=0.0, activ_f="gelu", final_norm="ln",
                                    input_f_size=64).to(dev).train()
    tr = FusionTrainStep(enc, comm="rccl")                   # no process group: the choice is accepted, nothing to create
    assert tr.bucket_comm is None
    with pytest.raises(ValueError):
        FusionTrainStep(enc, comm="mpi")


def test_layerwise_reducer_drives_the_own_communicator(monkeypatch):
    """The layer-wise reducer's CUDA branch with ``bucket_comm`` set: per-layer ``tf_allreduce_bucket`` calls on the communication stream,
    behind one event of the chain and one of the side stream, joined by ``finish()``.  One rank (sum over one rank = identity), so the
    step must end bit-identical to the same step without any exchange -- what is exercised is the ordering code, with real RCCL launches."""
    import os
    from cases import make_encoder_inputs, make_encoder_params
    from transfusion_amd.comm import BucketComm
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    from transfusion_amd.runner.trainer import FusionTrainStep
    monkeypatch.setenv("TF_FORCE_LAYERWISE", "1")
    dev = torch.device("cuda", 0)
    d, L, h, B, Nv, Nl = 64, 3, 4, 2, 20, 12
    params = make_encoder_params(77, d, L)
    x, lang, mask, gv, gl = make_encoder_inputs(77, B, Nv, Nl, d, [12, 5])
    t = lambda a: torch.from_numpy(a).to(dev)

    def run(with_comm):
        enc = CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 8192, d), num_layers=L,
                                        patch_dropout=0.0, num_heads=h, token_dropout=0.0, activ_f="gelu", final_norm="ln", input_f_size=d)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
        enc = enc.to(dev).train()
        tr = FusionTrainStep(enc, lr=1e-3, weight_decay=0.0, grad_clip=1.0)
        assert tr.layerwise is not None
        comm = None
        if with_comm:
            comm = BucketComm(1, 0, BucketComm.new_unique_id(), dev)
            tr.layerwise.live = True                     # take the exchange branch (the optimiser's 1 / world stays 1)
            tr.layerwise.bucket_comm = comm

        def loss_fn(m, batch):
            v, lo, _, _ = m(t(x), t(lang), t(mask))
            return (v.float() * t(gv)).sum() + (lo.float() * t(gl)).sum()

        for _ in range(3):
            tr.step([None], loss_fn)
        torch.cuda.synchronize()
        out = tr.flat.flat.clone()
        stats = None
        if comm is not None:
            stats = comm.stats()
            comm.close()
        return out, stats, tr

    ref, _, _ = run(False)
    got, stats, tr = run(True)
    assert stats["calls"] == 3 * L and stats["elems"] == 3 * tr.flat.grad.numel()     # every layer range, every step, nothing twice
    assert torch.equal(got, ref)


def test_close_leaves_no_communicator_thread_behind():
    """BucketComm.close() = ncclCommDestroy: the communicator's proxy / progress threads are joined.  (Round 5's GPU abort came from a
    native thread; the RCCL threads of this file's in-process communicators were the one native-thread source besides the HSA runtime.
    They were not the cause -- DESIGN.md "Round 6" -- and this file now runs at the END of the GPU suite, but the property is cheap to
    hold: after a first create / close has started whatever the library keeps for the process, a second one changes nothing.)"""
    import os
    from transfusion_amd.comm import BucketComm
    dev = torch.device("cuda", 0)
    threads = lambda: len(os.listdir("/proc/self/task"))

    def cycle():
        comm = BucketComm(1, 0, BucketComm.new_unique_id(), dev)
        g = torch.ones(4096, device=dev)
        comm.all_reduce_(g)
        torch.cuda.synchronize()
        during = threads()
        comm.close()
        return during

    cycle()                          # whatever the library starts once per process exists now ...
    cycle()                          # ... including what it only starts at a second communicator
    before = threads()
    during = [cycle() for _ in range(3)]
    after = threads()
    assert after <= before, (before, during, after)          # steady state: communicators come and go, the thread count does not grow
