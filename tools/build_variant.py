"""Build the library with extra hipcc flags into another file, for A/B runs with tools/ab_*.py (UPKIE_HIP_LIBRARY).
Usage: python tools/build_variant.py ab/libupkie_hip_x.so [-DUPKIE_... ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from upkie_amd import lib

out, flags = os.path.abspath(sys.argv[1]), sys.argv[2:]
os.makedirs(os.path.dirname(out), exist_ok=True)
lib.LIB_PATH = out
lib.HIPCC_FLAGS = lib.HIPCC_FLAGS + flags
print(lib.build(force=True))
