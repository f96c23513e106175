"""A/B of builds of the library on ONE box with tools/quick_bench.py (the fused Pendulum step at given batch sizes, optionally a forced
lane mapping), each build in its own process, interleaved. Usage: python tools/ab_quick.py [--lanes L] [--rounds N] --sizes 4096,262144 libA.so libB.so ..."""
import os
import subprocess
import sys

args = sys.argv[1:]


def take(flag, default):
    if flag in args:
        i = args.index(flag)
        v = args[i + 1]
        del args[i:i + 2]
        return v
    return default


lanes, rounds, sizes = take("--lanes", None), int(take("--rounds", "2")), take("--sizes", "4096").split(",")
tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quick_bench.py")
for r in range(rounds):
    for lib in args:
        env = dict(os.environ, UPKIE_HIP_LIBRARY=os.path.abspath(lib))
        if lanes:
            env["UPKIE_LANES_PER_ENV"] = lanes
        res = subprocess.run([sys.executable, tool] + sizes, env=env, capture_output=True, text=True)
        rows = [line.split() for line in res.stdout.splitlines() if line.startswith("B=")]
        print(f"round {r} {os.path.basename(lib):30s} " + "   ".join(f"B={row[1]}: {row[2]} us" if len(row) > 3 else str(row) for row in rows) + (res.stderr[-300:] if not rows else ""), flush=True)
