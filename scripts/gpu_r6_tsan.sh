#!/bin/bash
# Round 6 (as round 5): ThreadSanitizer over the HOST side -- the library's own host code (cwsl_gpu.hip rebuilt with -fsanitize=thread -fno-gpu-sanitize: its
# context mutex, life_mu, fetch tickets, per-receiver staging, batch stages, stats atomics) and the two threaded host programs (cwsl_gpu_realtime:
# one pusher thread per receiver / a batch pusher, clock, fetch threads; cwsl_gpu_skimmer: source threads, UDP), driven by their own GPU tests: the
# x8 real-time run with bit-identical sampled channels, the skimmer end to end, and the library's concurrency tests (tests/test_gpu_lifecycle.py)
# from Python threads with the runtime preloaded.  Built with ROCm's clang (gcc 11's libtsan dies on the box's address-space layout).
# The ROCm runtime itself (libamdhip64, libhsa-runtime64) is not instrumented: TSan cannot see its internal synchronisation and reports its
# internals against each other -- those reports are counted apart; a report with a frame in OUR sources is a finding.
# GPU AddressSanitizer / XNACK are not available on this pool and are not used.
O=$GRAFT_REPO_ROOT/gpurun_out; rm -rf $O/tsan; mkdir -p $O/tsan; cd $GRAFT_REPO_ROOT
B=cwsl_digi_amd/bin; L=$PWD/cwsl_digi_amd/lib/tsan; mkdir -p $L
RT=$(ls /opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
hipcc --offload-arch=gfx950 -O1 -g -ffp-contract=off -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-result -fno-slp-vectorize -fsanitize=thread -fno-gpu-sanitize \
      -shared-libsan -o $L/libcwslgpu.so cwsl_digi_amd/csrc/cwsl_gpu.hip -ldl 2>/dev/null || exit 1
for prog in skimmer realtime; do
  /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=thread -shared-libsan -Wall cwsl_digi_amd/csrc/host/${prog}_main.cpp -o $B/cwsl_gpu_${prog}_tsan -L$L -lcwslgpu \
      -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$(dirname $RT) -lpthread || exit 1
done
export CWSLG_SKIMMER_BIN=$PWD/$B/cwsl_gpu_skimmer_tsan CWSLG_REALTIME_BIN=$PWD/$B/cwsl_gpu_realtime_tsan
export TSAN_OPTIONS="halt_on_error=0 exitcode=0 second_deadlock_stack=1 history_size=4 log_path=$O/tsan/report"
{
echo "== host programs (TSan builds) through their GPU tests"
timeout 1500 python -m pytest tests/test_gpu_realtime.py tests/test_gpu_skimmer.py -q -m gpu 2>&1 | tail -4
echo "== the library's concurrency tests from Python threads, TSan library + runtime preloaded"
CWSLG_LIB=$L/libcwslgpu.so LD_PRELOAD=$RT timeout 1500 python -m pytest tests/test_gpu_lifecycle.py -q -m gpu -k "concurrent or atomic" 2>&1 | tail -4
python3 - <<'PY'
import glob, os, re, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "tsan")
txt = "".join(open(f, errors="replace").read() for f in sorted(glob.glob(O + "/report.*")))
OURS = re.compile(r"(cwsl_gpu\.hip|_main\.cpp|iq_source\.hpp|slot_clock\.hpp|skimmer_config\.hpp|spot_parse\.hpp|handoff\.hpp|host_dsp\.hpp|cwsl_gpu_shim\.hpp|\.inc):\d+")
def owner(frames):
    """Whose access is it: the first frame that is neither the sanitizer's interceptor nor the C++ runtime."""
    for f in frames:
        if "libclang_rt.tsan" in f or "libstdc++" in f or "libc.so" in f or "/tsan/rtl/" in f:
            continue
        if "libamdhip64" in f or "libhsa-runtime64" in f or "librccl" in f:
            return "rocm"
        return "ours" if OURS.search(f) else "other: " + f.strip()[:80]
    return "unknown"
kinds, ours = collections.Counter(), []
for r in txt.split("=================="):
    if "WARNING: ThreadSanitizer" not in r:
        continue
    first = r.split("WARNING: ThreadSanitizer: ")[1].splitlines()[0].split(" (pid")[0]
    stacks = re.findall(r"\n  (?:Write|Read|Previous write|Previous read|Atomic read|Atomic write|Previous atomic write|Previous atomic read)[^\n]*\n((?:    #\d[^\n]*\n)+)", r)
    who = sorted(owner(s.splitlines()) for s in stacks[:2])
    kinds[(first, " vs ".join(who))] += 1
    if who and all(w == "ours" for w in who):
        ours.append(r)
print("== ThreadSanitizer reports by kind (whose the two racing accesses are: `rocm` = inside libamdhip64 / libhsa-runtime64, which are not")
print("   instrumented -- an object handed to the runtime inside a HIP call of ours and touched later by the runtime's own threads)")
for k, v in kinds.most_common():
    print("%5d  %s | %s" % (v, k[0], k[1]))
print("== reports with BOTH racing accesses in our sources: %d" % len(ours))
for r in ours[:6]:
    print(r[:3500])
PY
} > $O/r6_tsan.txt 2>&1
cat $O/r6_tsan.txt
