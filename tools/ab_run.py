"""Bench every build_ab/libkmm_<name>.so (tools/ab_build.sh) with the same bench.py command, one after the other on
the same box, and print ms per step and per kernel.   python tools/ab_run.py [names...] [-- bench.py args]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
check = "--check" in args          # also run the radix parity tests with every build
args = [a for a in args if a != "--check"]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
names = args or sorted(os.path.basename(p)[7:-3] for p in glob.glob(os.path.join(ROOT, "build_ab", "libkmm_*.so")))
for name in names:
    env = dict(os.environ, KMM_LIB_PATH=os.path.join(ROOT, "build_ab", "libkmm_%s.so" % name))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--no-h2d-leg", "--no-records-host-leg"] + extra
    if check:
        t = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_radix.py"), "-x", "-q"],
                           env=env, capture_output=True, text=True, cwd=ROOT)
        print("%-18s tests: %s" % (name, t.stdout.strip().splitlines()[-1] if t.stdout.strip() else t.stderr[-300:]), flush=True)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        print("%-18s FAILED rc=%d %s" % (name, r.returncode, r.stderr[-400:]), flush=True)
        continue
    j = json.loads(line[-1])
    pk = {k: v.get("avg_ms") for k, v in j["roofline"].get("per_kernel", {}).items()}
    print("%-18s %8.1f G/s %7.3f ms  %s" % (name, j["value"] / 1e3 if j["unit"].startswith("M") else j["value"], j["ms_per_step"], pk), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ab_%s.json" % name), "w") as f:
        f.write(line[-1] + "\n")
