import cProfile, pstats, sys, os, io
sys.argv = ["probe_train_step.py", "detector", os.environ.get("B", "2"), "6"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pr = cProfile.Profile()
src = open(os.path.join(ROOT, "scripts", "probe_train_step.py")).read()
pr.enable()
exec(compile(src, "probe_train_step.py", "exec"), {"__name__": "__main__", "__file__": os.path.join(ROOT, "scripts", "probe_train_step.py")})
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[-6000:])
