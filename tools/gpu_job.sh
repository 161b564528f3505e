#!/bin/bash
# The ONE wrapper every GPU-box job of a session goes through:
#     /usr/local/graft/bin/gpurun --timeout T -- 'bash tools/gpu_job.sh <tag> [args...]'
# It sets up what rocprofv3 needs (cwd /tmp, TMPDIR), creates gpurun_out/<tag>/ (merged back by gpurun) and runs the job BODY
# tools/jobs/<tag>.sh with $O = that directory and $R = the repo root.  Job bodies are session scratch (what to measure in this
# call) and are not tracked (.gitignore: tools/jobs/); the summaries worth keeping are copied to profiles/ by hand.
# Helpers for bodies:  pmc <outdir> <counters...> -- <cmd...>   one rocprofv3 counter pass (counters only, no trace domains)
#                      stats <outdir> -- <cmd...>               rocprofv3 --kernel-trace --stats
set -u
tag=${1:?usage: gpu_job.sh <tag> [args...]}; shift
cd /tmp && export TMPDIR=/tmp
export R=${GRAFT_REPO_ROOT:-/root/repo}
export O=$R/gpurun_out/$tag
rm -rf "$O"; mkdir -p "$O"
pmc() { local d=$1; shift; local c=(); while [ "$1" != "--" ]; do c+=("$1"); shift; done; shift
        timeout 900 rocprofv3 --pmc "${c[@]}" --kernel-trace --output-format csv -d "$d" -- "$@" > "$d.log" 2>&1; }
stats() { local d=$1; shift; shift
          timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- "$@" > "$d.log" 2>&1; }
export -f pmc stats
cd "$R"
body=$R/tools/jobs/$tag.sh
[ -f "$body" ] || { echo "no job body $body"; exit 2; }
bash "$body" "$@" 2>&1 | tee "$O/job.log" | tail -n 200
# raw counter dumps are large: keep the summaries the body made, drop the per-dispatch CSVs
find "$O" -name "*_counter_collection.csv" -size +2M -delete 2>/dev/null
find "$O" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
exit 0
