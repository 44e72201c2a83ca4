"""Lab: isc_beam_select launched back to back (warm instruction cache) and between other kernels (as in a beam step), one
image x beam 5, V = 10 000, four state planes of 512 - run under rocprofv3 --kernel-trace and read the per-launch durations
(tools/select_lab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insenticap_model_amd import _lib, ops  # noqa: E402

DEV = torch.device('cuda:0')


def main():
    g = torch.Generator().manual_seed(3)
    n_img, beam, T, V, K, H = 1, 5, 20, 10000, 512, 512
    rows = n_img * beam
    tw = ops.rows_stats_tile(V)
    nt = (V + tw - 1) // tw
    cv = torch.randn(rows, nt, 8, generator=g).sort(dim=2, descending=True).values.to(DEV)
    ci = (torch.arange(nt)[None, :, None] * tw + torch.arange(8)[None, None, :]).expand(rows, nt, 8).to(torch.int32).contiguous().to(DEV)
    pm = cv[:, :, 0].contiguous()
    ps = torch.rand(rows, nt, generator=g).add(1).to(DEV)
    last = torch.randint(4, V, (rows,), generator=g).to(DEV)
    score = torch.randn(rows, generator=g).double().to(DEV)
    words = torch.randint(4, V, (rows, T), generator=g).to(DEV)
    length = torch.full((rows,), 3, dtype=torch.int32, device=DEV)
    st_in, st_out = torch.randn(4, rows, H, generator=g).to(DEV), torch.zeros(4, rows, H, device=DEV)
    out = dict(score=torch.zeros(rows, dtype=torch.float64, device=DEV), last=torch.zeros(rows, dtype=torch.int64, device=DEV),
               words=torch.zeros(rows, T, dtype=torch.int64, device=DEV), length=torch.zeros(rows, dtype=torch.int32, device=DEV),
               done=torch.zeros(n_img, dtype=torch.int32, device=DEV), src=torch.zeros(rows, dtype=torch.int64, device=DEV),
               live=torch.zeros(T + 1, dtype=torch.int32, device=DEV))
    a = _lib.BeamSelectArgs()
    a.n_img, a.beam, a.T, a.t, a.n_tile, a.V, a.eos_id = n_img, beam, T, 3, nt, V, 2
    a.part_max, a.part_sum, a.cand_val, a.cand_idx = pm.data_ptr(), ps.data_ptr(), cv.data_ptr(), ci.data_ptr()
    a.score_in, a.score_out, a.last_in, a.last_out = score.data_ptr(), out['score'].data_ptr(), last.data_ptr(), out['last'].data_ptr()
    a.words_in, a.words_out, a.len_in, a.len_out = words.data_ptr(), out['words'].data_ptr(), length.data_ptr(), out['length'].data_ptr()
    a.done, a.src_row, a.live = out['done'].data_ptr(), out['src'].data_ptr(), out['live'].data_ptr()
    a.state_in, a.state_out, a.state_planes, a.H = st_in.data_ptr(), st_out.data_ptr(), 4, H
    big = torch.randn(64 << 20, device=DEV)
    w = torch.randn(4096, 4096, device=DEV)
    for _ in range(3):
        ops.beam_select(a)
    torch.cuda.synchronize()
    for _ in range(20):                 # back to back
        ops.beam_select(a)
    torch.cuda.synchronize()
    for _ in range(20):                 # between other kernels (256 MB streamed, a GEMM)
        big.mul_(1.0001)
        (w @ w)
        ops.beam_select(a)
    torch.cuda.synchronize()
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    a.live_in = zero.data_ptr()         # the search has ended: the launch returns at once (the floor of a launch)
    for _ in range(20):
        big.mul_(1.0001)
        (w @ w)
        ops.beam_select(a)
    torch.cuda.synchronize()
    print('select_lab done')


if __name__ == '__main__':
    main()
