#!/bin/bash
# AddressSanitizer + UBSan run of the host-side plan compiler (no GPU needed): tests/native/plan_sanitize.cpp
cd "$(dirname "$0")/.." && g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Iopfgym_amd/csrc \
  tests/native/plan_sanitize.cpp opfgym_amd/csrc/plan.cpp -o /tmp/plan_sanitize && ASAN_OPTIONS=detect_leaks=1 /tmp/plan_sanitize
