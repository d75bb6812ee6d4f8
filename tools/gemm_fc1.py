#!/usr/bin/env python3
"""the ViT fc1 GEMM (59136 x 4352 x 1152, bias + GELU) six times: target of tools/pmc.sh"""
import os, sys
sys.argv = [sys.argv[0], "59136", "4352", "1152", "gelu"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_one.py")).read())
