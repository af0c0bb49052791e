cd $GRAFT_REPO_ROOT
for h in 1 0; do echo "H3=$h"; ADVOC_H3=$h timeout 600 python -m pytest tests/test_hip_model.py -q -m gpu -k side_stream 2>&1 | grep -E "AssertionError: \(|passed|failed" | head -5; done
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_hip_model.py::test_side_stream_weight_gradients_equal_serial_execution 2>&1 | tail -15
