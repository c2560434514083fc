"""The reference's own run (src/main.cpp: PiecewiseICP_4D_call(configuration_4d.txt, 0, 20, pairMode, 0.75) on its 20 scans,
kept as fixtures) through libpwicp.so: wall time of the call, second call of the process.  PWICP_TRACE=1 for the stages."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for rep in range(3):
    d = tempfile.mkdtemp()
    cfg = os.path.join(d, "cfg.txt")
    open(cfg, "w").write("string FolderFilePath1: %s\nstring FolderFilePath2: %s/\nbool isSetResSVsize (yes-1, no-0): 1\n"
                         "float PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\n"
                         "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\nbool isVisual (yes-1, no-0): 0"
                         % (os.path.join(ROOT, "tests", "golden", "inputs"), d))
    os.chdir(d)
    fd = os.dup(1); os.dup2(2, 1)                    # the entry point prints the reference's progress lines on stdout
    t0 = time.perf_counter()
    ok = P.PiecewiseICP_4D_call(cfg, 0, 20, mode, 0.75)
    t = time.perf_counter() - t0
    os.dup2(fd, 1); os.close(fd)
    print("call %d (pairMode %d): %s, %.3f s for 19 pairs" % (rep, mode, ok, t), flush=True)
