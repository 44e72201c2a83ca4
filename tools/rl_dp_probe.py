import os, sys
sys.path.insert(0, '.')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29658')
import torch, torch.distributed as dist
import bench
bench.load_product()
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
for B in (64, 512):
    r = bench.bench_rl(dev, iters=20, B=B)
    print('no group  B=%d: %.1f ms  cider %.1f' % (B, r['ms_per_iter'], r['cider_ms_per_iter']), flush=True)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
for B in (64, 512):
    r = bench.bench_rl(dev, iters=20, B=B)
    print('one-rank group B=%d: %.1f ms  cider %.1f  all-reduces %d' % (B, r['ms_per_iter'], r['cider_ms_per_iter'], r['grad_arena_all_reduces']), flush=True)
dist.destroy_process_group()
