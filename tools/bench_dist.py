"""GPU box: the standalone pixel->prototype distance kernel alone in a process (for the PMC traffic passes)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
import bench
print(json.dumps(bench.bench_distance_kernel(16, 768, torch.device("cuda", 0))))
