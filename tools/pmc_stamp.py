#!/usr/bin/env python3
"""Prints the kernel-source digest bench.py compares `roofline.traffic` records against (run on the GPU box next to the
PMC passes, so that the record names the sources the counters were actually measured on)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_summary import kernel_source_sha  # noqa: E402
print(kernel_source_sha())
