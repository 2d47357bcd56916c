#!/usr/bin/env python3
"""Build and run tools/threads_rate.c against libmicro_aes_hip_128.so (the drop-in API, host buffers):
calls per second of 1..16 concurrent host threads.  Usage: threads_rate.py [calls_per_thread] [bytes]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "micro-aes_amd", "lib")
exe = "/tmp/uaes_threads_rate"
subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "threads_rate.c"),
                "-o", exe, "-L" + lib, "-lmicro_aes_hip_128", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib",
                "-lpthread"], check=True)
sys.exit(subprocess.run([exe] + sys.argv[1:]).returncode)
