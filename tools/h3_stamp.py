#!/usr/bin/env python3
"""Reads the s_memtime stamps of the diagnostic library (tools/h3_stamp.sh): share of setup / main loop / epilogue in the
vocabulary form of gemm_h3_kernel and in the LSTM / linear forms of gemm_h3x_kernel over greedy roll-outs at B = 16384 (the
headline workload) and 4096."""
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
NAMES = ('gemm_h3_kernel<vocab>', 'gemm_h3x_kernel<lstm> K=1024 (att-LSTM)', 'gemm_h3x_kernel<lstm> K=1536 (lang-LSTM)',
         'gemm_h3x_kernel<linear>')
for B in (16384, 4096):
    inputs, _ = bench.device_inputs(B, 100, dev)
    with torch.no_grad():
        for _ in range(2):
            cap(*inputs, bench.T, 1, mode='rl')
        torch.cuda.synchronize()
        out = (C.c_uint64 * 16)()
        lib.isc_debug_h3_stamps(out, 1)
        for _ in range(3):
            cap(*inputs, bench.T, 1, mode='rl')
        torch.cuda.synchronize()
        lib.isc_debug_h3_stamps(out, 1)
    for s in range(4):
        waves, setup, loop, epi = [int(x) for x in out[4 * s:4 * s + 4]]
        if not waves:
            continue
        tot = max(1, setup + loop + epi)
        print('B=%d: %d waves of %s: stamp ticks per wave setup %.0f  main loop %.0f  epilogue %.0f  ->  shares '
              '%.3f / %.3f / %.3f' % (B, waves, NAMES[s], setup / waves, loop / waves, epi / waves,
                                      setup / tot, loop / tot, epi / tot))
