"""A/B of builds of the library on ONE box on bench.py's secondary blocks -- the C5 share (device-drawn push schedule, law
inside the launch, both laws), C3 (UpkieBaseVelocity + MPC balancer, 16384 envs) and the public VecEnv.step loop: each build
in its own process, interleaved. Usage: python tools/ab_c5.py libA.so libB.so [--rounds N]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1]))))
import bench
out = []
for law in ("velocity", "torque"):
    r = bench.secondary_c5_share(law, steps=1000, warmup=200, census_steps=0)
    out.append(f"{law} {r['us_per_step']:.2f}")
r = bench.secondary_c3(steps=1000, warmup=200)
out.append(f"c3 {r['us_per_step']:.2f}")
r = bench.vec_env_api(steps=1000, warmup=200)
out.append(f"vec_env next_step {r['next_step']['python_loop_us_per_env_step']:.2f} same_step {r['same_step']['python_loop_us_per_env_step']:.2f}")
print("  ".join(out))
'''
args = sys.argv[1:]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds")
    rounds = int(args[i + 1])
    del args[i:i + 2]
for r in range(rounds):
    for lib in args:
        env = dict(os.environ, UPKIE_HIP_LIBRARY=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print(f"round {r} {os.path.basename(lib):32s} us/step: {res.stdout.strip() or res.stderr[-400:]}", flush=True)
