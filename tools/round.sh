#!/bin/bash
# One script for the GPU-box runs of a round (replaces the per-round one-offs).  Run through gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/round.sh <what> [tag] [args...]'
# Everything it writes goes to gpurun_out/$ROUND/ (scratch; copy what is to be judged into profiles/).
#   tests [tag] [pytest args]     the GPU suite (or the files / -k expression given)
#   check_ab [tag]                the check tests, then the check half at 64 and at 576 slices (decoder A/B between builds)
#   bench [tag] [bench args]      one bench.py line -> bench_<tag>.json
#   line [tag]                    the driver's default line (python bench.py)
#   stats [tag] [bench args]      rocprofv3 --kernel-trace --stats of a bench run -> stats_<tag>/
#   pmc [tag] [bench args]        the PMC passes of tools/profile_pmc.sh over a bench run
#   cumask                         rocprofv3 modes against a process with a CU-masked stream (tools/rocprof_cumask_repro.hip)
#   checkprof [tag]                check half: per-launch HIP events as timed (partitioned), then rocprofv3 --stats + FETCH/WRITE_SIZE passes unpartitioned
#   encprof [tag]                  encode half: rocprofv3 --stats, FETCH/WRITE_SIZE passes, SQ instruction counters at batch 336
#   trace [tag] [bench args]       rocprofv3 --kernel-trace as a timeline: where k_resolve / k_rangecode stand still (TRACE_MODE=check: the check half's launches)
#   damage [tag] [from] [to]       the damage soak of route C: both binaries' verdicts on damaged files, seeds from..to-1
#   refresh [tag]                  tests + encprof + checkprof + the 576-slice and 8K passes + the driver's line, on one box (then tools/adopt_profiles.sh)
#   sweep [tag] "ENV=.." ...       one bench line per environment setting with the TIMING build -> sweep_<tag>.jsonl, e.g. the floor of round 4:
#                                  sweep floor "" RCGPU_EXP_SKIP_RC=1 "RCGPU_EXP_SKIP_RC=1 RCGPU_EXP_STATES_L2=1" RCGPU_EXP_STATES_L2=1 RCGPU_RC_SPAN=64
ROUND=${ROUND:-r06}
WHAT=${1:-tests}; TAG=${2:-x}; shift 2 2>/dev/null
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
OUT=gpurun_out/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline") or {}
    print(f, d.get("metric"), d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"),
          "identical", (d.get("config") or {}).get("all_frames_identical_to_source"))
PY
}
case $WHAT in
tests)
    [ $# -gt 0 ] || set -- tests
    timeout 2400 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 | tee $OUT/tests_$TAG.log ;;
check_ab)
    timeout 900 python -m pytest tests/test_gpu_check.py tests/test_gpu_stages.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/tests_$TAG.log
    timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" > $OUT/check_$TAG.json 2> $OUT/check_$TAG.err || tail -3 $OUT/check_$TAG.err
    timeout 600 python bench.py --mode check --steps 2 --warmup 1 --legs "" --slices 576 --check-batch 256 > $OUT/check576_$TAG.json 2> $OUT/check576_$TAG.err || tail -3 $OUT/check576_$TAG.err
    show $OUT/check_$TAG.json $OUT/check576_$TAG.json ;;
bench)
    timeout 1500 python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || tail -5 $OUT/bench_$TAG.err
    show $OUT/bench_$TAG.json ;;
line)
    timeout 1500 python bench.py > $OUT/line_$TAG.json 2> $OUT/line_$TAG.err || tail -5 $OUT/line_$TAG.err
    show $OUT/line_$TAG.json ;;
stats)
    cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/stats_$TAG" -o s -- python "$OLDPWD/bench.py" "$@" > "$OLDPWD/$OUT/stats_$TAG.json" 2> "$OLDPWD/$OUT/stats_$TAG.err"; cd "$OLDPWD"
    python tools/rocprof_summary.py $OUT/stats_$TAG 2>/dev/null | head -30 ;;
pmc)
    bash tools/profile_pmc.sh $OUT/pmc_$TAG "$@" ;;
cumask)
    # which rocprofv3 modes survive a process that has made a CU-masked stream (tools/rocprof_cumask_repro.hip)
    B=$PWD/tools/bin/rocprof_cumask_repro; R=$PWD/$OUT/rocprof_cumask.txt; : > $R
    [ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/rocprof_cumask_repro.hip -o $B
    cd /tmp
    for masked in 1 2 0; do
        echo "== second stream $([ $masked = 1 ] && echo CU-masked || ([ $masked = 2 ] && echo 'CU-masked and alive at exit' || echo plain))" >> $R
        echo "alone: $($B $masked 2>&1 | tail -1) (exit $?)" >> $R
        for mode in "--kernel-trace" "--kernel-trace --stats" "--pmc GRBM_GUI_ACTIVE" "--kernel-trace --pmc GRBM_GUI_ACTIVE" "--memory-copy-trace"; do
            rm -rf /tmp/rp_cm; timeout 120 rocprofv3 $mode -d /tmp/rp_cm -o x -- $B $masked > /tmp/rp_cm.log 2>&1; rc=$?
            echo "rocprofv3 $mode: exit $rc$([ $rc = 139 ] && echo ' (SIGSEGV)'); $(grep -c . /tmp/rp_cm.log) lines of output; last: $(grep -v '^$' /tmp/rp_cm.log | tail -1 | cut -c1-160)" >> $R
        done
    done
    rocprofv3 --version 2>&1 | head -3 >> $R
    cat $R ;;
oomrepro)
    B=$PWD/tools/bin/oom_cumask_repro; [ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/oom_cumask_repro.hip -o $B
    for m in 0 1 2; do $B $m 2>&1 | grep -v amdgpu.ids | tail -1; echo "  mode $m -> exit ${PIPESTATUS[0]}"; done | tee $OUT/oom_cumask.txt ;;
cumaskpy)
    R=$PWD/$OUT/rocprof_cumask_py.txt; : > $R; P=$PWD/tools/rocprof_cumask_repro.py
    cd /tmp
    for v in lib lib+free torch torch+lib torch+lib+free torch+mask; do
        rm -rf /tmp/rp_py; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_py -o x -- python $P $v > /tmp/rp_py.log 2>&1; rc=$?
        echo "rocprofv3 --kernel-trace --stats -- python tools/rocprof_cumask_repro.py $v: exit $rc$([ $rc = 139 ] && echo ' (SIGSEGV)'); said: $(grep '^ok' /tmp/rp_py.log | tail -1); trace written: $(find /tmp/rp_py -name '*.db' 2>/dev/null | wc -l) database(s); $(grep -c __cxa_finalize /tmp/rp_py.log) x __cxa_finalize in the stack" >> $R
    done
    cat $R ;;
checkprof)
    # the check half as the driver times it (hash on CUs of its own): HIP events per launch -> profiles-ready JSON; then the same under rocprofv3 --stats
    # (its trace survives since bench.py gives the CU-masked hash streams back before it ends), and the PMC passes for k_dec_slices' HBM bytes
    timeout 600 python bench.py --mode check --steps 3 --warmup 1 --legs "" --check-offsets 0,4096,2097152,1052672,0,33554432,16781312 --check-profile-out $OUT/check_partitioned_$TAG.json > $OUT/checkprof_$TAG.json 2> $OUT/checkprof_$TAG.err || tail -3 $OUT/checkprof_$TAG.err
    timeout 600 python bench.py --mode check --steps 3 --warmup 1 --legs "" --slices 576 --check-batch 512 --check-profile-out $OUT/check576_partitioned_$TAG.json > $OUT/checkprof576_$TAG.json 2> $OUT/checkprof576_$TAG.err || tail -3 $OUT/checkprof576_$TAG.err
    show $OUT/checkprof_$TAG.json $OUT/checkprof576_$TAG.json
    PASS_TIMEOUT=500 bash tools/profile_check.sh ${ROUND}${TAG} > $OUT/profile_check_$TAG.log 2>&1; tail -12 $OUT/profile_check_$TAG.log | cut -c1-220
    cp gpurun_out/summary/${ROUND}${TAG}_check_* $OUT/ 2>/dev/null ;;
encprof)
    # the encode half: rocprofv3 --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE passes (tools/profile_enc.sh), then the SQ instruction counters at batch 336
    PASS_TIMEOUT=500 bash tools/profile_enc.sh ${ROUND}${TAG} > $OUT/profile_enc_$TAG.log 2>&1; tail -25 $OUT/profile_enc_$TAG.log | cut -c1-220
    cp gpurun_out/prof/${ROUND}${TAG}_* $OUT/ 2>/dev/null
    RCGPU_BENCH_BATCH=336 PMC_GROUPS="3" PMC_PASS_TIMEOUT=500 bash tools/profile_pmc.sh ${ROUND}${TAG}_sq > $OUT/profile_sq_$TAG.log 2>&1; tail -12 $OUT/profile_sq_$TAG.log | cut -c1-200
    cp gpurun_out/pmc/${ROUND}${TAG}_sq.csv $OUT/ 2>/dev/null ;;
trace)
    # rocprofv3 --kernel-trace of a bench run as a timeline (tools/kernel_trace.py): TRACE_MODE=check for the check half
    cd /tmp; rm -rf /tmp/rp_tr; timeout 600 rocprofv3 --kernel-trace -d /tmp/rp_tr -o t -- python "$OLDPWD/bench.py" ${@:---steps 4 --warmup 1 --legs= --no-verify} > "$OLDPWD/$OUT/trace_$TAG.log" 2>&1; cd "$OLDPWD"
    python tools/kernel_trace.py "$(find /tmp/rp_tr -name '*.db' | head -1)" ${TRACE_MODE:-encode} | tee $OUT/trace_$TAG.txt | tail -40 ;;
sweep)
    # A/B over environment settings with the TIMING build (the shipped library reads no measuring switch): one bench line per argument
    # ("A=1 B=2" "A=3" ...; "" = no setting) -> $OUT/sweep_<tag>.jsonl.  SWEEP_ARGS = extra bench.py arguments.
    : > $OUT/sweep_$TAG.jsonl
    for cfg in "$@"; do
        env RCGPU_LIB=rawcooked_amd/librcgpu_timing.so RCGPU_BENCH_TIMING_BUILD=1 $cfg timeout 400 python bench.py --steps ${SWEEP_STEPS:-3} --warmup 1 --legs "" ${SWEEP_ARGS:-} > /tmp/line.json 2>/tmp/line.err || tail -3 /tmp/line.err
        python - "$cfg" /tmp/line.json >> $OUT/sweep_$TAG.jsonl <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    print(json.dumps({"env": sys.argv[1], "frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "frames_per_step": c.get("frames_per_step_per_gpu"),
                      "kernel_ms_per_step": {k: round(v, 1) for k, v in r.get("kernel_ms_per_step", {}).items()},
                      "verified_vs_oracle": c.get("verified_vs_oracle"), "device_error_flags": c.get("device_error_flags"), "run_on": c.get("run_on")}))
except Exception as e:
    print(json.dumps({"env": sys.argv[1], "error": str(e)}))
PY
        tail -1 $OUT/sweep_$TAG.jsonl
    done ;;
damage)
    # the linked binary's verdict against the unmodified reference's on damaged Matroska files (tests/test_gpu_e2e.py, ~65 s a seed, eight damaged
    # copies each): damage [tag] [first seed] [one past the last]; what both binaries said goes to $OUT/damage_<seed>_<kind>.txt on a mismatch
    RCGPU_DAMAGE_DUMP=$PWD/$OUT/damage RCGPU_SOAK_DAMAGE_FROM=${1:-0} RCGPU_SOAK_DAMAGE=${2:-8} timeout 2400 python -m pytest tests/test_gpu_e2e.py -k "says_what_the_reference_says" -x -q -m gpu 2>&1 | tail -15 | tee $OUT/damage_$TAG.log ;;
refresh)
    # the end-of-round set on the final tree, one box: GPU suite, every pass profiles/traffic.json is made from (64 slices, 576, 8K, the check half),
    # the driver's line.  Then, at home: bash tools/adopt_profiles.sh $TAG
    bash $0 tests $TAG; grep -q "passed" $OUT/tests_$TAG.log && ! grep -q "failed\|error" $OUT/tests_$TAG.log || { echo "round.sh refresh: the suite is not green, nothing else is run"; exit 1; }
    bash $0 encprof $TAG; bash $0 checkprof $TAG
    PASS_TIMEOUT=500 bash tools/profile_enc.sh ${ROUND}${TAG}_576 --slices 576 --batch 40 > $OUT/profile_enc576_$TAG.log 2>&1; tail -8 $OUT/profile_enc576_$TAG.log | cut -c1-200
    PASS_TIMEOUT=500 bash tools/profile_enc.sh ${ROUND}${TAG}_8k --width 8192 --height 4320 --slices 576 --batch 80 > $OUT/profile_enc8k_$TAG.log 2>&1; tail -8 $OUT/profile_enc8k_$TAG.log | cut -c1-200
    cp gpurun_out/prof/${ROUND}${TAG}_* $OUT/ 2>/dev/null
    bash tools/adopt_profiles.sh $TAG $ROUND | tail -6      # profiles/traffic.json of THIS tree, here on the box, so that the line below carries it
    bash $0 bench $TAG --gpus 1 --steps 20 --warmup 5 ;;
*)  echo "round.sh: unknown '$WHAT'"; exit 2 ;;
esac
