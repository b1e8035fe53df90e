"""The reference's evaluation call (all rays of an 800x800 image from host memory) through OctreeRender_trilinear_fast: bench.dropin_eval_ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
dev = torch.device("cuda", 0)
field, params, aabb = bench.build_field(dev)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in bench.dropin_eval_ms(field, dev, 800, 800).items()})
