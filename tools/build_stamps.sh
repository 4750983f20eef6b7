#!/bin/bash
# Diagnostic build with in-kernel phase stamps (never shipped; see csrc/gp_fit_fused.hip SCAML_STAMPS)
set -e
cd "$(dirname "$0")/.."
P="scalable-meta-learning-with-gaussian-processes_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DSCAML_STAMPS $P/csrc/gp_fit_fused.hip -o $P/lib/libscaml_hip_stamps.so
