"""Compile the C++ host mirror's frame-loop driver against libwsfluid.so (g++, no HIP needed)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(HERE, "frame_loop")


def build_host():
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", HERE,
           os.path.join(HERE, "frame_loop.cpp"), "-o", OUT, "-L", PKG, "-lwsfluid", "-pthread", "-Wl,-rpath," + PKG]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_host())
