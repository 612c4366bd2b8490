#!/bin/bash
# Every `-m gpu` test file the emulator can run, under the AddressSanitizer build of the emulator (tests/simt/build.py, SIMT_SANITIZE=1):
# one line per file -> profiles/r06_simt_asan.txt.  CPU only; an hour or two on 8 cores.  usage: bash tools/simt_asan_full.sh [file ...]
cd "$(dirname "$0")/.."
LEVEL=${SIMT_SANITIZE:-1}      # 1 = AddressSanitizer, 2 = + UndefinedBehaviorSanitizer (profiles/r06_simt_ubsan.txt)
LIB=$(SIMT_SANITIZE=$LEVEL python tests/simt/build.py | tail -n 1) || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export SUBLINEAR_HIP_LIB=$LIB SIMT_ALLOW=1 SIMT_THREADS=${SIMT_THREADS:-8} SIMT_FAKE_TORCH=2 SL_RCCL_LIB=$(dirname $LIB)/librccl.so.1 SL_COMM_TIMEOUT_MS=600000
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=0
O=profiles/r06_simt_asan.txt; [ "$LEVEL" = 2 ] && O=profiles/r06_simt_ubsan.txt
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
FILES=${@:-$(ls tests/test_gpu_*.py | grep -v "test_gpu_bench\|test_gpu_multi_device\|test_gpu_dist_abi\|test_gpu_fullsize\|test_gpu_acl")}
[ $# -eq 0 ] && { echo "# the -m gpu files under the AddressSanitizer build of the SIMT emulator (tools/simt_asan_full.sh), HEAD $(git rev-parse --short HEAD)" > $O; echo "# deselected: full-size instances and programs linked against the real library (NOT_HERE of tests/test_simt_emulated.py), the 2^21-walk stride test" >> $O; }
DES="--deselect tests/test_gpu_parity.py::test_c3_full_size_properties --deselect tests/test_gpu_parity.py::test_cpp_host_mirror --deselect tests/test_gpu_pagerank.py::test_c4_full_size_pagerank_queries --deselect tests/test_gpu_order_any.py::test_order_any_headline_instance_sampled --deselect tests/test_gpu_cli.py::test_c_program_solves_through_the_abi --deselect tests/test_gpu_cli.py::test_javascript_surface_on_gpu --deselect tests/test_gpu_degenerate.py::test_slice_pointers_that_do_not_match_the_row_lengths_are_noticed_and_rebuilt --deselect tests/test_gpu_fuzz.py::test_nothing_relies_on_fresh_device_memory_being_zero --deselect tests/test_gpu_walk.py::test_block_stride_shrinks_beyond_the_generators_period --deselect tests/test_gpu_panels.py::test_seven_million_short_rows_many_thin_panels --deselect tests/test_gpu_parity.py::test_c2_full_solve_1m"
for f in $FILES; do
  T0=$(date +%s)
  LD_PRELOAD=$RT timeout 5000 python -m pytest -q -m gpu -p no:cacheprovider --timeout 2400 $f $DES > /tmp/asan_one.log 2>&1; RC=$?
  printf '%-36s rc %d  %5ds  %s  sanitizer reports: %s\n' "$f" $RC $(( $(date +%s) - T0 )) "$(grep -E ' passed| failed| error' /tmp/asan_one.log | tail -n 1)" "$(grep -c 'ERROR: AddressSanitizer\|runtime error:' /tmp/asan_one.log)" | tee -a $O
  [ $RC -ne 0 ] && { grep -A 25 'ERROR: AddressSanitizer\|runtime error:' /tmp/asan_one.log | head -40 >> $O; tail -n 15 /tmp/asan_one.log >> $O; }
done
