#!/usr/bin/env python3
"""Reads the s_memtime stamps of the diagnostic library (tools/h3_stamp.sh): share of setup / main loop / epilogue in the
vocabulary form of gemm_h3_kernel over greedy roll-outs at B = 4096 (the headline workload)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
from insenticap_model_amd import Captioner, _lib, synth

lib = _lib.load()
lib.isc_debug_h3_stamps.restype = C.c_int
lib.isc_debug_h3_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
dev = torch.device('cuda:0')
cap = Captioner(synth.make_idx2word(bench.V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(bench.V, synth.DEFAULT_SETTINGS).items()})
cap.to(dev).eval()
for B in (4096, 128):
    inputs, _ = bench.device_inputs(B, 100, dev)
    with torch.no_grad():
        for _ in range(3):
            cap(*inputs, bench.T, 1, mode='rl')
        torch.cuda.synchronize()
        out = (C.c_uint64 * 4)()
        lib.isc_debug_h3_stamps(out, 1)
        for _ in range(5):
            cap(*inputs, bench.T, 1, mode='rl')
        torch.cuda.synchronize()
        lib.isc_debug_h3_stamps(out, 1)
    waves, setup, loop, epi = [int(x) for x in out]
    tot = max(1, setup + loop + epi)
    print('B=%d: %d waves of gemm_h3_kernel<vocab>: cycles per wave setup %.0f  main loop %.0f  epilogue %.0f  ->  shares '
          '%.3f / %.3f / %.3f' % (B, waves, setup / max(1, waves), loop / max(1, waves), epi / max(1, waves),
                                  setup / tot, loop / tot, epi / tot))
