"""Child program of tests/test_gpu_dp.py::test_rccl_all_reduce_in_a_fresh_process - NOT a test module.

Started as a fresh process (nothing has touched the GPU before `init_from_env`), it builds a one-rank process group with
backend "nccl" (= RCCL on ROCm), runs two `xe_train_step`s of the tiny captioner with a gradient arena (a one-rank group
takes every data-parallel branch of the step: the 3-float count all-reduce, the loss shares, the arena all-reduce and the
loss-statistics all-reduce - `dp.distributed()`), and prints one
JSON line with what the parent asserts on: the backend, that librccl is mapped into the process, how many collectives
the arena issued, and that the parameters moved.  NCCL_DEBUG=INFO output (RCCL's own log of the AllReduce calls) goes to
stdout next to it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', sys.argv[1] if len(sys.argv) > 1 else '29533')
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('NCCL_DEBUG', 'INFO')
os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,COLL')

import torch                                            # noqa: E402
import torch.distributed as dist                        # noqa: E402

from insenticap_model_amd import Captioner, dp, synth   # noqa: E402
from insenticap_model_amd.train import xe_train_step    # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    V, st = 64, synth.TINY_SETTINGS
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=9).items()})
    cap.to('cuda:0').eval()
    dp.broadcast_parameters(cap)                        # world 1: returns at once
    arena = dp.GradArena(cap.parameters())
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
    d = synth.make_inputs(8, V, st, regions=6, seq_len=8, seed=31)
    s = synth.make_inputs(4, V, st, regions=6, seq_len=8, seed=32)
    t = torch.from_numpy
    fact = (None, t(d['fc_feats']), t(d['att_feats']), (t(d['captions']), d['lengths']), t(d['cpt_words']))
    scs = ((t(s['captions']), s['lengths']), t(s['cpt_words']), t(s['senti_words']), t(s['senti_labels']))
    before = {k: v.detach().clone() for k, v in cap.state_dict().items()}
    losses = []
    for _ in range(2):
        out = xe_train_step(cap, optim, xe_crit, da_crit, fact, t(d['senti_labels']), scs, 0.0, 0.1, arena=arena)
        losses.append(float(out['all_loss']))
    # a second, explicit collective whose result is checkable: sum over 1 rank == identity, bit for bit
    probe = torch.arange(arena.flat.numel(), dtype=torch.float32, device='cuda:0') * 0.5
    arena.flat.copy_(probe)
    arena.all_reduce()
    torch.cuda.synchronize()
    identity = bool(torch.equal(arena.flat, probe))
    all_reduces_eager = dp.COLLECTIVES
    # the same iteration from HIP graphs under the process group (train_graph.XETrainGraph: forward + backward and
    # clamp + Adam are graphs, the three collectives run between them): five steps against an eager twin
    from insenticap_model_amd.train_graph import XETrainGraph

    def twin():
        m = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, st)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=9).items()})
        m.to('cuda:0').eval()
        return m, dp.GradArena(m.parameters())
    # (graphs run one chain per unroll, eager steps the merged chain - autograd_pair.use_pair: the twin takes the
    # graph's form, and the flat exchange the graph object issues between its two graphs)
    os.environ['ISC_PAIR_UNROLLS'] = '0'
    ref, ref_arena = twin()
    ro, rx, rd = ref.get_optim_criterion(4e-4)
    for _ in range(5):
        xe_train_step(ref, ro, rx, rd, fact, t(d['senti_labels']), scs, 0.0, 0.1, arena=ref_arena)
    gm, g_arena = twin()
    go, gx, gd = gm.get_optim_criterion(4e-4)
    tg = XETrainGraph(gm, go, gx, gd, grad_clip=0.1, arena=g_arena, warmup=2)
    c0 = dp.COLLECTIVES
    # The backend's watchdog thread polls its work events every ~100 ms.  Keep the capture window open for 0.4 s (a host
    # sleep inside the captured phase): had the all-reduce of the loss shares issued just before still been on the
    # watchdog's list, its poll inside the window would raise hipErrorCapturedEvent in the watchdog and abort this
    # process (seen at the full model size, where captures are that long by themselves); ops.graph_capture finishes
    # all queued work and lets one watchdog pass go by before it opens a capture.
    import time
    from insenticap_model_amd import ops
    real_refresh = ops.refresh_weight_planes

    def slow_refresh(epoch_before):
        if torch.cuda.is_current_stream_capturing():
            time.sleep(0.4)
        return real_refresh(epoch_before)
    ops.refresh_weight_planes = slow_refresh
    for _ in range(5):
        tg.step(fact, t(d['senti_labels']), scs, 0.0)
    ops.refresh_weight_planes = real_refresh
    torch.cuda.synchronize()
    graph = dict(replays=tg.replays, eager=tg.eager_steps, all_reduces=dp.COLLECTIVES - c0,
                 arena_collectives=g_arena.collectives,
                 equal=all(bool(torch.equal(a, b)) for a, b in zip(ref.state_dict().values(), gm.state_dict().values())))
    with open('/proc/self/maps') as f:
        maps = f.read()
    moved = sum(int(not torch.equal(before[k], v)) for k, v in cap.state_dict().items())
    print('RCCL_CHILD ' + json.dumps(dict(
        backend=dist.get_backend(), rccl_mapped=('librccl' in maps), collectives=arena.collectives,
        all_reduces=all_reduces_eager, graph=graph,
        arena_bytes=arena.nbytes, losses=losses, identity=identity, moved=moved,
        nccl_version=list(torch.cuda.nccl.version()))), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
