#!/bin/bash
timeout 600 python tools/gpu_cfg5_levels.py 2>&1 | grep -v amdgpu.ids
