#!/bin/bash
# Builds the C++ host-API test binary (g++; links the product library and, as the checker, the oracle).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/bin"
for t in test_icp test_model_estimation; do
g++ -O2 -std=c++17 -ffp-contract=off -I"$ROOT/include" "$HERE/$t.cpp" -o "$HERE/bin/$t" \
    -L"$ROOT/cilantro_amd/lib" -lcilantro_hip -L"$ROOT/oracle" -loracle \
    -Wl,-rpath,"$ROOT/cilantro_amd/lib" -Wl,-rpath,"$ROOT/oracle" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
done
# host-side check of the fast paths in csrc/solve.hpp (hipcc: the header pulls in the HIP runtime API; runs without a GPU)
/opt/rocm/bin/hipcc -O2 -std=c++17 -ffp-contract=off --offload-arch=gfx950 "$HERE/test_solve.cpp" -o "$HERE/bin/test_solve"
# tests/cpp/tie_order_host.hpp (the host restatement of the reference's index build: the cross-check of csrc/tie_build.hip) as a host library
g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC -pthread "$HERE/tie_order_shim.cpp" -o "$HERE/bin/libtie_order_shim.so"
# host-only PLY round-trip helper (no GPU library needed)
g++ -O2 -std=c++17 -I"$ROOT/include" "$HERE/test_ply.cpp" -o "$HERE/bin/test_ply"
# the example program (compile check; run it on a GPU box with a PLY file)
g++ -O2 -std=c++17 -I"$ROOT/include" "$ROOT/examples/rigid_icp.cpp" -o "$HERE/bin/example_rigid_icp" \
    -L"$ROOT/cilantro_amd/lib" -lcilantro_hip -Wl,-rpath,"$ROOT/cilantro_amd/lib" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
