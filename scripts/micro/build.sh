#!/bin/bash
# builds the standalone micro-benchmarks against the in-tree library
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/micro/mix_roofline.hip -o scripts/micro/mix_roofline -Lml-qem_amd/csrc -lmlqem_hip -Wl,-rpath,'$ORIGIN/../../ml-qem_amd/csrc'
