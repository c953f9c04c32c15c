# profiles/<tag>_* from gpurun_out/prof_r6 (tools/dev/profile_r6.sh): usage tools/dev/summarize_r6.sh r06_v1
# gpurun MERGES a call's gpurun_out/ into the local one: clear gpurun_out/prof_r6 locally BEFORE every profile_r6.sh call, or the
# summariser averages the counter files of several runs (it refuses below when a pass directory holds more than one run)
set -e
t=$1; s=gpurun_out/prof_r6
for d in $s/*stats $s/*pmc_*; do [ -d "$d" ] || continue; n=$(find $d -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | wc -l); if [ "$n" -gt 1 ]; then echo "more than one run in $d: clear $s and profile again"; exit 1; fi; done
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
