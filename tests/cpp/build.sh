#!/bin/bash
# Builds the C++ host-API test binary (g++; links the product library and, as the checker, the oracle).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/bin"
g++ -O2 -std=c++17 -ffp-contract=off -I"$ROOT/include" "$HERE/test_icp.cpp" -o "$HERE/bin/test_icp" \
    -L"$ROOT/cilantro_amd/lib" -lcilantro_hip -L"$ROOT/oracle" -loracle \
    -Wl,-rpath,"$ROOT/cilantro_amd/lib" -Wl,-rpath,"$ROOT/oracle" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
