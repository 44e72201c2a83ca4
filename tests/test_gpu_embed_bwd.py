"""Embedding + ReLU backward (deterministic scatter-add): the indexed entry point isc_embed_relu_bwd_ws (count / offsets /
segments / sorted per-id sums) against the scanning entry point it replaces - bit for bit - and against a plain fp64
scatter-add, over the layouts the captioner uses: plain token rows, rows_per_grad > 1 with a scale (concept words),
the sentiment-word layout with its <PAD> prefix, a keep-mask, a skipped padding id, and ids hot enough that one
segment exceeds the 2048-entry LDS list (windowed path).  pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch

from insenticap_model_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _call(entry, emb, ids, ids_stride, n_rows, rows_per_grad, pad_first, pad_id, dout, scale, mask, mask_scale, demb,
          skip_id, ws=None):
    lib = ops._lib.load()
    V, W = emb.shape
    args = [emb.data_ptr(), V, W, ids.data_ptr(), ids_stride, n_rows, rows_per_grad, pad_first, pad_id,
            dout.data_ptr(), scale, ops.ptr(mask), mask_scale, demb.data_ptr(), skip_id]
    if entry == 'ws':
        args += [ws.data_ptr(), ws.numel() * 4]
        ops.check(lib.isc_embed_relu_bwd_ws(*args, ops.stream()), 'isc_embed_relu_bwd_ws')
    else:
        ops.check(lib.isc_embed_relu_bwd(*args, ops.stream()), 'isc_embed_relu_bwd')


@pytest.mark.parametrize('n_rows,V,W,rows_per_grad,pad_first,masked,skip', [
    (5000, 300, 512, 1, 0, False, -1),       # hot ids: segments of ~17 ... and, below, > 2048
    (20480, 10000, 512, 1, 0, False, 0),     # an XE step's fed tokens, <PAD> row skipped
    (6000, 7, 96, 1, 0, True, -1),           # seven ids: every segment ~860, with a keep-mask
    (9000, 3, 64, 1, 0, False, -1),          # three ids: segments of ~3000 > the LDS list (windowed)
    (2560, 500, 512, 5, 0, False, -1),       # concept words: 5 rows per gradient row, scale 1/5
    (1200 * 11, 400, 128, 1, 11, True, 2),   # sentiment-word layout: row 0 of every image is <PAD> (= skipped id 2)
])
def test_indexed_embedding_backward_equals_the_scanning_one(n_rows, V, W, rows_per_grad, pad_first, masked, skip):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n_rows + V)
    emb = (torch.rand(V, W, generator=g) - 0.3).to(dev)                  # ~30 % of the entries fail the ReLU test
    if pad_first:
        B = n_rows // pad_first
        ids = torch.randint(0, V, (B, pad_first - 1), generator=g).to(dev)
        pad_id = skip
    else:
        ids = torch.randint(0, V, (n_rows,), generator=g).to(dev)
        pad_id = 0
    dout = torch.randn(n_rows // rows_per_grad, W, generator=g).to(dev)
    mask = (torch.rand(n_rows, W, generator=g) > 0.4).to(torch.uint8).to(dev) if masked else None
    scale, mask_scale = (1.0 / rows_per_grad), 1.7
    base = torch.randn(V, W, generator=g).to(dev)
    ws = torch.empty(4 * V + 64 + n_rows + 16, dtype=torch.int32, device=dev)
    out = {}
    for entry in ('scan', 'ws', 'ws'):
        demb = base.clone()
        _call(entry, emb, ids, 1, n_rows, rows_per_grad, pad_first, pad_id, dout, scale, mask, mask_scale, demb, skip, ws)
        torch.cuda.synchronize()
        out.setdefault(entry, []).append(demb.cpu())
    assert torch.equal(out['ws'][0], out['scan'][0])                     # same summation order: bit-identical
    assert torch.equal(out['ws'][0], out['ws'][1])                       # and repeatable
    # fp64 reference
    if pad_first:
        full = torch.cat([torch.full((B, 1), pad_id, dtype=torch.int64), ids.cpu()], dim=1).reshape(-1)
    else:
        full = ids.cpu()
    grad = dout.double().cpu().repeat_interleave(rows_per_grad, dim=0) * scale
    if masked:
        grad = grad * mask.cpu().double() * mask_scale
    ref = torch.zeros(V, W, dtype=torch.float64)
    keep = full != skip
    ref.index_add_(0, full[keep], grad[keep])
    ref = base.double().cpu() + ref * (emb.cpu() > 0).double()
    np.testing.assert_allclose(out['ws'][0].double().numpy(), ref.numpy(), atol=2e-3 * max(1.0, n_rows / V) ** 0.5,
                               rtol=1e-5)


def test_small_calls_fall_back_to_the_scanning_kernel():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    emb = torch.rand(50, 64, generator=g).to(dev)
    ids = torch.randint(0, 50, (100,), generator=g).to(dev)
    dout = torch.randn(100, 64, generator=g).to(dev)
    a, b = torch.zeros(50, 64, device=dev), torch.zeros(50, 64, device=dev)
    ops.embed_relu_bwd(emb, ids, dout, a, 100)                           # through ops: workspace given, n < 1024
    _call('scan', emb, ids, 1, 100, 1, 0, 0, dout, 1.0, None, 1.0, b, -1)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_position_index_inside_a_captured_graph_with_new_ids_every_replay():
    """Training graphs replay the indexed backward with other token ids every time (sampled roll-outs, scheduled sampling),
    several calls per graph on one workspace.  Its counters were once cleared by hipMemsetAsync: captured as a memset
    node that did not always clear them on replay - counts piled up, emb_fill_kernel wrote past the list and
    emb_accumulate_kernel read slots nobody had written (GPU memory faults in the RL training graph under a process
    group).  They are cleared by a kernel now: replays with fresh ids equal the eager launches bit for bit."""
    g = torch.Generator().manual_seed(11)
    V, W, n = 3000, 96, 5120
    emb = (torch.rand(V, W, generator=g) - 0.3).to(DEV)
    douts = [torch.randn(n, W, generator=g).to(DEV) for _ in range(3)]
    ids = [torch.zeros(n, dtype=torch.int64, device=DEV) for _ in range(3)]
    outs = [torch.zeros(V, W, device=DEV) for _ in range(3)]

    def draw(k):
        # skewed: a few hundred distinct ids, some of them thousands of times (window mode), <PAD> skipped in call 0
        hot = torch.randint(0, 40, (n,), generator=g)
        cold = torch.randint(0, V, (n,), generator=g)
        pick = torch.rand(n, generator=g) < (0.3 + 0.2 * k)
        return torch.where(pick, hot, cold)

    def launches():
        for o in outs:
            o.zero_()
        ops.embed_relu_bwd(emb, ids[0], douts[0], outs[0], n, skip_id=0)
        ops.embed_relu_bwd(emb, ids[1], douts[1], outs[1], n, scale=0.5)
        ops.embed_relu_bwd(emb, ids[2], douts[2], outs[2], n)
    for k in range(3):
        ids[k].copy_(draw(k))
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        launches()                                      # warm (allocates the stream's workspace outside the capture)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            launches()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    for rep in range(5):
        for k in range(3):
            ids[k].copy_(draw(k + rep))
        graph.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        with torch.cuda.stream(st):
            launches()
        torch.cuda.synchronize()
        for a, b in zip(got, outs):
            assert torch.equal(a, b), rep
