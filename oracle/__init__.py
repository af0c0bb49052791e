"""CPU oracle for the AdVoc hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(paarthneekhara/advoc, TF1 / lws / librosa) used as the *checker* for the HIP
kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``advoc_amd``) never does: it fails loudly if the HIP library is missing.

Pinning status ("Oracle"):
  * spectral_np  -- pinned against the reference's own known-answer constants
                    (tests/test_spectral.py:33-46,74-76) and the r9y9 fixture.  Its iSTFT /
                    Griffin-Lim part reproduces the reference's Griffin-Lim constants
                    (tests/test_spectral.py:200-203) to 0.1 % with scipy's resampler standing in
                    for librosa's (8-decimal pin NOT reproducible).
  * lws_np       -- PARITY UNPINNED: lws 1.2 (third-party C++) is not part of /root/reference; the published
                    LWS algorithm restated.  The one reachable reference number (run_lws from the true complex
                    spectrogram, tests/test_spectral.py:190,207: 4.24e-4) is met in order of magnitude (5.7e-4).
  * audioio      -- pinned against outputs of the reference's advoc/audioio.py
                    imported in the build container (tests/golden/make_golden.py).
  * advoc_torch  -- PARITY UNPINNED: the reference holds no test or golden
                    output for the conv stack, losses or optimiser; the oracle
                    is a torch-CPU restatement of TF1 semantics cross-checked
                    against closed-form micro cases (tests/test_oracle_conv.py).
  * loader_np    -- PARITY UNPINNED (no reference test); restates
                    advoc/loader.py:133-186 and tf.contrib.signal.frame.
  * melspecgan_torch -- STRUCTURE PINNED (r3): equals, to float64 round-off, the TensorFlow-written inference graph the
                    reference holds (models/melspecgan/infer.meta, decoded into tests/golden/melspecgan_graph.json and
                    evaluated op by op by tests/tf_graph_interp.py; variable table, BN epsilon / mode, transposed-conv
                    strides / padding / output shapes, output mapping).  Trained weights are not reachable: no numeric
                    golden output exists.  The same file pins TF's Conv2DBackpropInput SAME rule, which
                    advoc_torch.gen_deconv shares.
"""
