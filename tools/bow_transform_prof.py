"""Profiling target: bench.py's bow_transform leg alone (k = 10, L = 6 synthetic vocabulary, 64 x 2000 resident descriptors, 50 launches
of k_bow_transform).  usage (GPU box): rocprofv3 --kernel-trace --stats -- python3 tools/bow_transform_prof.py   (or --pmc FETCH_SIZE / WRITE_SIZE)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
out = bench.bow_bench(torch, torch.device("cuda", 0), batch_only=True)   # only the 64 x 2000 launches: the trace's average is theirs
print(json.dumps({k: out[k] for k in ("launch_ms_avg", "launch_ms_median", "verified", "descriptors_per_s")}))
