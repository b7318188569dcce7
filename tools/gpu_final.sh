# the -m gpu suite in the three arithmetic modes, then the profile collection and the counter passes
cd $GRAFT_REPO_ROOT
bash tools/gpu_test_all_modes.sh ${1:-final}_tests
bash tools/collect_profiles.sh ${1:-final}
bash tools/gpu_prof_counters.sh ${1:-final}/counters > /dev/null 2>&1
