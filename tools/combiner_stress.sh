#!/bin/bash
# Runs ON THE GPU BOX: the drop-in door under concurrency WITH a check of the results (tools/configs0_mt.c, check mode): state k is fed the
# deterministic signal k % 7 and keeps a checksum of every output sample and VAD value.  Within a run all states of one signal must agree;
# across runs -- one thread over 7 states, then up to 128 threads over up to 1,024 states, pools of 1,024 / 256 / 64 rows, followers
# sleeping or spinning -- the seven checksums must be the same (every run: 1,000 frames per state).  A frame lost, repeated or delivered
# to the wrong row by the combiner shows either way.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
( cd "$R" && gcc -O2 -Iinclude tools/configs0_mt.c -o /tmp/configs0_mt -Lrnnoise_amd -l:librnnoise_amd.so -Wl,-rpath,$R/rnnoise_amd -lpthread ) || exit 2
python -c "import lzma;open('/tmp/default.blob','wb').write(lzma.decompress(open('$R/tests/golden/default.blob.xz','rb').read()))"
echo "# tools/combiner_stress.sh: $(nproc) CPUs visible, cgroup quota $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
rc=0
run() {  # threads states [env...]
  local t=$1 s=$2; shift 2
  echo "## $t threads, $s states $*"
  env "$@" timeout 300 /tmp/configs0_mt /tmp/default.blob $t 900 $s 1 2>&1 | tee -a /tmp/stress.out || rc=1
}
: > /tmp/stress.out
run 1 7
run 16 16
run 64 64
run 64 1024
run 128 256
run 96 300
run 64 1024 RNNOISE_AMD_POOL_ROWS=256
run 64 256 RNNOISE_AMD_POOL_ROWS=64
run 32 70 RNNOISE_AMD_COMBINE_WAKE_EARLY_US=0
run 16 16 RNNOISE_AMD_COMBINE=0
n=$(grep "^check:" /tmp/stress.out | sed 's/.*checksums//' | sort -u | wc -l)
bad=$(grep -c MISMATCH /tmp/stress.out)
echo "# runs with a mismatch inside: $bad; distinct checksum sets over all runs: $n (1 = every run produced the same seven streams)"
[ "$bad" = 0 ] && [ "$n" = 1 ] && [ $rc = 0 ] || exit 3
