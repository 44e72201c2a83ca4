import sys, torch
sys.path.insert(0, '.')
import bench
bench.load_product()
print(bench.bench_epoch_loops(torch.device('cuda:0')))
