"""un-fused gate-apply micro-benchmark only (bench.py's gate_apply leg) -- for A/B of kernel variants via env vars"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.gate_apply_microbench(torch.device("cuda:0"), n=16)
print(os.environ.get("BORNVI_GATE1Q_VARIANT", "0"), os.environ.get("BORNVI_GATE1Q_GRID", "32"), json.dumps(r["gbs"]), r["frac_of_hbm_peak"])
