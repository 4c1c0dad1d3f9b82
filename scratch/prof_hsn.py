import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--workload", "hsn", "--arch", "vgg16", "--batch", "16", "--steps", "5", "--warmup", "1"]
sys.path.insert(0, os.getcwd())
import bench
pr = cProfile.Profile(); pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
