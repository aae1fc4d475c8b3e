export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03a_gputests.txt; cat gpurun_out/r03a_gputests.txt
( timeout 300 bash tests/manual/wgrad_atomic_slabs.sh 2>&1 | tail -30 ) > gpurun_out/r03a_wgrad_atomic.txt; cat gpurun_out/r03a_wgrad_atomic.txt
( timeout 300 bash tests/manual/lstm_four_sequence_sweeps.sh 2>&1 | tail -40 ) > gpurun_out/r03a_lstm4.txt; cat gpurun_out/r03a_lstm4.txt
