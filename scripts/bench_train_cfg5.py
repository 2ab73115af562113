"""cfg-5 (box corrector, R101-FPN, 2 x 800x1333) training step alone: the `train_cfg5_r101` object of bench.py, for profiler runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
c = bench._setup()
print(json.dumps(bench.train_leg(c, int(os.environ.get("STEPS", "5")), 2, 2, "cfg5")))
