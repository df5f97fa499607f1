"""Time of one 1024 x 1024 x 2179 cosine cost matrix launch (bench.py: pairwise_roofline), optionally per env setting."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "strotss-tensorflow_amd")]
import torch
import bench
print(os.environ.get("STROTSS_X3_XCD_BLOCK", "default"), bench.pairwise_roofline(torch.device("cuda", 0), iters=200))
