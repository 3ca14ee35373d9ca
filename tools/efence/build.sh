#!/bin/bash
# builds the fence allocator (test infrastructure; see efence.cpp)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -shared -fPIC -O1 efence.cpp -o libefence.so
echo "$(pwd)/libefence.so"
