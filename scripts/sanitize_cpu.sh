#!/bin/bash
# UndefinedBehaviorSanitizer over the CPU-side C / C++ of this repository (GPU sanitizers are not available on the pool): the oracle (gcc, -fsanitize=undefined,
# non-recovering) under tests/test_oracle_cpu.py, and the compiler-scheduled lane bodies as the host emulation compiles them (clang, trap mode: undefined behaviour
# ends the process) under tests/test_emul_cpu.py + tests/test_fused_subgroup_cpu.py. The regular builds are put back afterwards.
set -e
cd "$(dirname "$0")/.."
cp oracle/_build/libbls_oracle.so /tmp/libbls_oracle.so.keep
cp tests/host_emul/libmbls_emul.so /tmp/libmbls_emul.so.keep
restore() { cp /tmp/libbls_oracle.so.keep oracle/_build/libbls_oracle.so; cp /tmp/libmbls_emul.so.keep tests/host_emul/libmbls_emul.so; touch oracle/_build/libbls_oracle.so tests/host_emul/libmbls_emul.so; }
trap restore EXIT
gcc -O1 -g -fPIC -std=gnu11 -fsanitize=undefined -fno-sanitize-recover=undefined -shared -o oracle/_build/libbls_oracle.so oracle/bls_oracle.c -lpthread
UBSAN_OPTIONS=halt_on_error=1 python -m pytest tests/test_oracle_cpu.py -x -q
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=undefined -fsanitize-trap=undefined -o tests/host_emul/libmbls_emul.so tests/host_emul/mbls_emul.cpp
touch tests/host_emul/libmbls_emul.so
python -m pytest tests/test_emul_cpu.py tests/test_fused_subgroup_cpu.py -x -q
echo "UBSan: clean"
