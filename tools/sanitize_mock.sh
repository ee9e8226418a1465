#!/bin/bash
# The device code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only): the product sources compiled against the mock HIP
# runtime (tests/hostsim) with -fsanitize=address,undefined, then the mock-runtime tests -- or any command given -- run on that build.
# Device buffers are plain heap blocks there, so a kernel that reads or writes past an arena, a slot or a result array is reported
# with a stack trace instead of passing silently as it would on the GPU.  ~3.5 min to build, tests run ~1.5x slower than usual.
#   tools/sanitize_mock.sh                          # tests/test_hostsim.py test_dp_fuzz.py test_opt_fuzz.py test_cli.py test_host_finalize.py
#   tools/sanitize_mock.sh python tools/mock_fuzz_campaign.py --minutes 30
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${BWA_AMD_SAN_DIR:-/tmp/bwa_amd_san}; mkdir -p $OUT
LIB=$OUT/libbwagpu_hostsim_san.so
if [ ! -e $LIB ] || [ -n "$(find $ROOT/bwa_amd/csrc $ROOT/tests/hostsim $ROOT/include -newer $LIB \( -name '*.h' -o -name '*.hip' -o -name '*.cpp' \) | head -1)" ]; then
  g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -x c++ -I $ROOT/tests/hostsim \
      $ROOT/bwa_amd/csrc/bwagpu.hip $ROOT/bwa_amd/csrc/bwagpu_index.hip $ROOT/tests/hostsim/mock_globals.cpp -o $LIB
fi
HOSTSRC=$ROOT/bwa_amd/csrc/host
if [ ! -e $OUT/bwa-amd-sim-san ] || [ $LIB -nt $OUT/bwa-amd-sim-san ] || [ -n "$(find $HOSTSRC -newer $OUT/bwa-amd-sim-san \( -name '*.h' -o -name '*.cpp' \) | head -1)" ]; then
  # the command line (reader, encoder, device, finalize and writer threads) and the host finalize library, same flags
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -DBWAGPU_CLI_TEST_HOOKS $HOSTSRC/*.cpp -o $OUT/bwa-amd-sim-san \
      -L$OUT -lbwagpu_hostsim_san -Wl,-rpath,$OUT -lz -lpthread
  g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer $(ls $HOSTSRC/*.cpp | grep -v main_) -o $OUT/libbwamem_host_san.so -lpthread
fi
export BWA_AMD_SIM_CLI=$OUT/bwa-amd-sim-san BWA_AMD_HOST_LIB=$OUT/libbwamem_host_san.so
rm -f $OUT/asan.* $OUT/ubsan.*
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:log_path=$OUT/asan
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$OUT/ubsan
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) BWA_AMD_HOSTSIM_LIB=$LIB
cd $ROOT
if [ $# -gt 0 ]; then "$@"; else python -m pytest tests/test_hostsim.py tests/test_dp_fuzz.py tests/test_opt_fuzz.py tests/test_cli.py tests/test_host_finalize.py -q -m "not gpu" -p no:cacheprovider; fi
rc=$?
ls $OUT/asan.* $OUT/ubsan.* 2>/dev/null && { echo "SANITIZER REPORTS under $OUT"; exit 1; }
echo "no sanitizer reports"; exit $rc
