# GPU session r4d: the tests that failed in r4c + the new ones, then the round's artefact set
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "attention or bf16_flavour or sample_then_decode or bench_two_ranks or training_loop or checkpoint or weights_reloaded" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -8 $O/tests.log
timeout 3000 bash tools/round_artifacts.sh r04 > $O/artifacts.log 2>&1
tail -5 $O/artifacts.log
ls gpurun_out/r04 gpurun_out | head -60
