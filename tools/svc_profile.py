#!/usr/bin/env python3
"""The bounded config-4 (SVC) sample of bench.py on its own, for rocprofv3 (kernel trace / PMC passes):
   rocprofv3 --kernel-trace --stats ... -- python tools/svc_profile.py [mesh]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
import pylabfea_amd as FE  # noqa: E402
from pylabfea_amd import _lib  # noqa: E402
print(json.dumps(bench.svc_sample(FE, _lib, int(sys.argv[1]) if len(sys.argv) > 1 else 128)))
