# profiles/<tag>_* from gpurun_out/prof_r6 (tools/dev/profile_r6.sh): usage tools/dev/summarize_r6.sh r06_v1
set -e
t=$1; s=gpurun_out/prof_r6
python tools/summarize_profile.py $s profiles/$t "MPC02 batch=1024" "" > /dev/null
python tools/summarize_profile.py $s profiles/${t}_soc "MPC02-SOC batch=1024" soc_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_tile "dense-front batch=512" tile_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_b512 "MPC02 batch=512" b512_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_b4096 "MPC02 batch=4096" b4096_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_afiro "lp_afiro batch=256" afiro_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_bandm "lp_bandm batch=256" bandm_ > /dev/null
python tools/summarize_profile.py $s profiles/${t}_25fv47 "lp_25fv47 batch=256" fv47_ > /dev/null
tail -1 $s/bench_plain.json > profiles/${t}_bench.json
[ -s $s/bench_multi8.json ] && tail -1 $s/bench_multi8.json > profiles/${t}_bench_multi8.json
ls profiles/${t}_*
