cd $GRAFT_REPO_ROOT
LIB=before ROUNDS=2 bash tools/env_ab.sh hplean2 "before-default|" "before-lean0|RNNOISE_AMD_HP_LEAN=0" "before-pipe1|RNNOISE_AMD_PIPE=1" "before-pipe1-lean0|RNNOISE_AMD_PIPE=1 RNNOISE_AMD_HP_LEAN=0"
LIB=xring ROUNDS=2 bash tools/env_ab.sh hplean2 "xring-default|" "xring-lean0|RNNOISE_AMD_HP_LEAN=0" "xring-pipe1|RNNOISE_AMD_PIPE=1" "xring-pipe1-lean0|RNNOISE_AMD_PIPE=1 RNNOISE_AMD_HP_LEAN=0"
LIB=before ROUNDS=1 bash tools/env_ab.sh hplean2 "before-default|" "before-lean0|RNNOISE_AMD_HP_LEAN=0"
LIB=xring ROUNDS=1 bash tools/env_ab.sh hplean2 "xring-default|" "xring-lean0|RNNOISE_AMD_HP_LEAN=0"
LIB=before ROUNDS=2 BENCH_ARGS="--streams 16384" bash tools/env_ab.sh hplean2 "before-16384|"
LIB=xring ROUNDS=2 BENCH_ARGS="--streams 16384" bash tools/env_ab.sh hplean2 "xring-16384|"
LIB=before ROUNDS=2 BENCH_ARGS="--streams 32768 --model little" bash tools/env_ab.sh hplean2 "before-little|" "before-little-lean0|RNNOISE_AMD_HP_LEAN=0"
LIB=xring ROUNDS=2 BENCH_ARGS="--streams 32768 --model little" bash tools/env_ab.sh hplean2 "xring-little|" "xring-little-lean0|RNNOISE_AMD_HP_LEAN=0"
