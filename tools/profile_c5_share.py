"""The C5 share under both servo-level laws, default contact model, for `rocprofv3 --kernel-trace --stats`: the eight-lane Servos
kernel (`step_kernel_octet<4, true, false, false, false>`, on-device policy) is the only kernel of the loop; its average duration
is what `secondary.c5_share_*` report per env.step().
Usage (GPU box): cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o c5 -- python tools/profile_c5_share.py [torque|velocity]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

for law in sys.argv[1:] or ["torque"]:
    r = bench.secondary_c5_share(law, census_steps=0)
    print(law, round(r["us_per_step"], 2), "us per env.step()")
