"""CPU oracle for the CTR hot path of wangruichens/recsys.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recsys_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED (by the reference): the reference repository contains no
tests, fixtures or golden vectors, and its arithmetic lives in TensorFlow 1.x,
which is neither vendored nor installable in this environment (SURVEY.md
section 8c).  The restatement below therefore follows the reference scripts
(cited file:line, paths relative to /root/reference) plus the TF-1.13/1.14
semantics they silently invoke.  It is pinned by the
known-answer tests of SURVEY.md Appendix B (FarmHash Fingerprint64 KATs from
upstream TF's string_to_hash_bucket tests, CRC-32C / TFRecord framing KATs,
bucketize table, TF-Adam first-step values, closed-form tiny cases) and by
fp64 finite-difference gradient checks, all in ``tests/``.

WHERE EACH TF-1.13 SEMANTIC LIVES (round 4): every behaviour of SURVEY.md Appendix A that the reference
scripts invoke silently, the upstream TensorFlow r1.13 source it was restated from (tensorflow/python/... unless said
otherwise), and the oracle line that implements it -- so that a reader WITH TensorFlow sources can check the whole restatement
in one sitting.  Nothing here was executed against TensorFlow (not installable in this environment): "parity unpinned".

  A-#  semantic (reference call site)                              upstream r1.13 file :: function                         oracle
  ---  ----------------------------------------------------------  ------------------------------------------------------  -----------------------------
  A-1  input_layer concatenates columns sorted by column.name,     feature_column/feature_column.py ::                     criteo.py field_table()   
       not in list order (fm/fm.py:117-118)                        _internal_input_layer (sorted(..., key=lambda x: x.name))  (:38), row_offsets() (:52)
  A-2  string -> id = Fingerprint64(bytes) % buckets; int64 keys   core/kernels/string_to_hash_bucket_op.h (Fingerprint64 =  hashing.py fingerprint64(),
       formatted as decimal strings first (fm/fm.py:89,              core/platform/fingerprint.h -> farmhashna::Hash64);       hashing.py hash_bucket()
       deepfm/deepfm.py:41)                                        feature_column.py :: _HashedCategoricalColumn
  A-3  bucketized_column buckets the NORMALISED value log(x+s)     feature_column.py :: _BucketizedColumn._transform_feature  criteo.py bucketize() (:57)
       with upper_bound; NaN -> last bucket (fm/fm.py:76-79)       -> core/kernels/bucketize_op.cc (std::upper_bound)
  A-4  embedding_column: mean combiner, truncated-normal(1/sqrt D)  feature_column.py :: _EmbeddingColumn;                    models.py gather_fm_fwd(),
       init; gradient = IndexedSlices over the unique ids           ops/embedding_ops.py :: safe_embedding_lookup_sparse      init.py trunc_normal()
  A-5  AdamOptimizer on IndexedSlices is NOT lazy: m, v decay and   training/adam.py :: _apply_sparse_shared (assign m*beta1,  nn.py AdamTF1.apply_sparse()
       var moves over the WHOLE variable every step; epsilon-hat;   scatter_add, assign_sub over var); ApplyAdam kernel:      (:127), apply_dense() (:119),
       dense variables through the fused ApplyAdam (fm/fm.py:162)   core/kernels/training_ops.cc :: ApplyAdam                 alpha() (:110)
  A-6  every input_layer(...) call opens a fresh variable scope ->  feature_column.py :: _internal_input_layer                 models.py XDeepFM: P['tables']
       xDeepFM owns TWO table sets (xdeepfm/xdeepfm.py:128,185)     (variable_scope(None, default_name='input_layer'))        feeds CIN, P['tables2'] the DNN (:238,266)
  A-7  tf.layers.dense glorot-uniform / zeros; get_variable         layers/core.py :: Dense; ops/init_ops.py ::               init.py glorot_uniform(),
       default initializer glorot-uniform (CIN filters); glorot     GlorotUniform / GlorotNormal (_compute_fans)              glorot_normal()
       normal over 1-D shapes (dcn/dcn.py:139-140)
  A-8  batch_normalization(training=True) never updates the moving  layers/normalization.py :: BatchNormalization.call (adds   nn.py bn_train_fwd() (:33),   
       statistics (no UPDATE_OPS dependency): EVAL uses mean 0 /    the updates to UPDATE_OPS; nobody runs them)              bn_eval_fwd() (:56)    
       var 1 (deepfm/deepfm.py:106,142-143)
  A-9  dropout: inverted scaling, TRAIN only (fm/fm.py:20)          layers/core.py :: Dropout -> nn_ops.py :: dropout          nn.py dropout_fwd() (:64)
  A-11 tf.metrics.auc: 200 thresholds, pred > t, epsilon 1e-6,      ops/metrics_impl.py :: auc, _confusion_matrix_at_          nn.py StreamingAUC
       trapezoid; accuracy on round-half-even (fm/fm.py:150-153)    thresholds, accuracy
  A-12 MirroredStrategy: loss scaled by 1/num_replicas, gradients   contrib/distribute/python/mirrored_strategy.py;            tests/test_dist_gloo.py
       summed, IndexedSlices concatenated (fm/fm.py:184-194)        training/optimizer.py :: _scale_loss, distribute/          (backward(dz / world), :45)
                                                                    cross_device_ops.py :: aggregate_tensors_or_indexed_slices
  A-13 conv1d with a width-1 filter = per-position matmul; Z is    ops/nn_ops.py :: conv1d                                    models.py XDeepFM CIN forward
       flattened f-major (xdeepfm/xdeepfm.py:145-169)
  A-14 TFRecord framing, masked CRC-32C, tf.train.Example wire      core/lib/io/record_writer.cc, core/lib/hash/crc32c.h       tfrecord.py
       format (fm/fm.py:100-112)                                    (Mask), core/example/example.proto / feature.proto
"""
