#!/bin/bash
cd $GRAFT_REPO_ROOT
rocm-smi --showmemorypartition --showcomputepartition --showperflevel --showpower --showmaxpower > gpurun_out/smi_info.log 2>&1
rocminfo > gpurun_out/rocminfo.log 2>&1
python3 - > gpurun_out/devinfo.log 2>&1 <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p)
print("clock_rate", getattr(p, "clock_rate", None), "mem clock", getattr(p, "memory_clock_rate", None), "bus", getattr(p, "memory_bus_width", None))
PY
