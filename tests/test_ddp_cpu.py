"""Data-parallel path on CPU with the gloo backend, world_size 2: the flat-buffer gradient all-reduce of
transfusion_amd.runner.trainer equals the single-process gradient of the concatenated batch, parameters that
never receive gradients are excluded statically, and bucket boundaries do not matter."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 3)
        self.heatmap_token = torch.nn.Parameter(torch.randn(1, 1, 6))      # never used, like the reference's
        self.frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfusion_amd.runner.trainer import DataParallelReducer, FlatParams
    torch.manual_seed(123)
    data = torch.randn(8, 6)
    model = Toy()
    flat = FlatParams(model)
    assert "heatmap_token" not in flat.names and "frozen" not in flat.names
    red = DataParallelReducer(flat.grad, bucket_mb=1e-4)        # ~26 floats per bucket: several buckets
    assert red.world == world and len(red.buckets) > 1
    shard = data[rank * 4:(rank + 1) * 4]
    model(shard).pow(2).sum().backward()                         # autograd accumulates into the flat views
    for h in red.all_reduce(async_op=True):
        h.wait()
    if rank == 0:
        torch.save({"grad": flat.grad.clone(), "names": flat.names, "slices": [(n, o, k) for n, _, o, k in flat.slices]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_flat_allreduce(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(123)
    data = torch.randn(8, 6)
    ref = Toy()
    ref(data).pow(2).sum().backward()
    named = dict(ref.named_parameters())
    for n, off, k in got["slices"]:
        torch.testing.assert_close(got["grad"][off:off + k], named[n].grad.reshape(-1), rtol=1e-5, atol=1e-6)


def test_single_process_reducer_is_noop():
    from transfusion_amd.runner.trainer import DataParallelReducer
    g = torch.ones(10)
    r = DataParallelReducer(g)
    assert r.world == 1 and r.all_reduce() == [] and torch.equal(g, torch.ones(10))


def test_layerwise_ranges_cover_flat_buffer_once():
    """The per-layer all-reduce slices (top layer + final norm first, layer 0 + kind embeddings last) tile the flat
    gradient buffer exactly: nothing reduced twice, nothing forgotten."""
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    from transfusion_amd.runner.trainer import FlatParams, LayerwiseReducer
    pe = PositionalEmbeddingLayer("sin1d", 64, 32)
    enc = CrossTransformerModuleBox(no_patches=64, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=3, patch_dropout=0.1,
                                    num_heads=2, fforward_multiplier=2, token_dropout=0.1, back_to_img_fn="regroup", activ_f="gelu",
                                    final_norm="ln", input_f_size=32)
    flat = FlatParams(enc)
    red = LayerwiseReducer(flat)
    assert red.num_layers == 3 and red.world == 1
    spans = sorted(red.ranges.values())
    assert spans[0][0] == 0 and spans[-1][1] == max(off + n for _, _, off, n in flat.slices)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    # every parameter slice lies in exactly one range, final norm with the top layer, kind embeddings with layer 0
    where = {}
    for name, _, off, num in flat.slices:
        hits = [l for l, (lo, hi) in red.ranges.items() if lo <= off and off + num <= hi]
        assert len(hits) == 1, name
        where[name] = hits[0]
    assert where["final_norm_layer.weight"] == 2 and where["image_kind_embedding"] == 0
    assert where["t_encoder.layers.1.linear1.weight"] == 1
