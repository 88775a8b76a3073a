/*
 * dzn.h — C ABI of libdzn_hip.so, the MI355X (gfx950) engine for the DiariZen
 * sliding-window inference hot path.
 *
 * Plain C, no torch types: every buffer is a raw pointer + sizes.  The caller owns
 * all I/O buffers (for PyTorch-ROCm these are tensor.data_ptr()); the library owns
 * weights and workspace.  All entry points return 0 on success or a negative
 * DZN_E_* code; the message is available through dzn_last_error().  Nothing throws
 * across the boundary.
 *
 * What each entry point replaces in the reference (paths relative to the
 * BUTSpeechFIT/DiariZen tree, PA/ = pyannote-audio/pyannote/audio/):
 *
 *   dzn_create + dzn_load_tensor + dzn_finalize_weights
 *       <- PA/core/model.py:360-369   instantiate(config["model"]["path"], args) ;
 *                                      model.load_state_dict(torch.load(ckpt))
 *          diarizen/models/eend/model_wavlm_conformer.py:26-76 (Model.__init__)
 *          PA/models/embedding/wespeaker/__init__.py:207-233 (WeSpeakerResNet34)
 *   dzn_segment_forward
 *       <- PA/core/inference.py:215       self.model(chunks.to(device))
 *          diarizen/models/eend/model_wavlm_conformer.py:238-264 (Model.forward)
 *          PA/core/inference.py:226 + PA/utils/powerset.py:103-128
 *                                      (Powerset.to_multilabel, soft=False)
 *   dzn_embed_forward
 *       <- PA/pipelines/speaker_verification.py:693-705 (embedding __call__)
 *          PA/models/embedding/wespeaker/__init__.py:190-204 (fbank -> resnet)
 *          PA/models/embedding/wespeaker/resnet.py:344-376 ; PA/models/blocks/pooling.py:44-131
 *   dzn_destroy            <- python object lifetime
 *
 * Threading: one handle per device; calls on one handle are not re-entrant.  Calls
 * only ENQUEUE work on the given HIP stream; the caller synchronises (matching the
 * reference's single-threaded, one-stream use).  One exception, once per geometry: the
 * FIRST forward at a new window length builds that length's index tables (relative-position
 * bias table, positional-conv and ResNet row-offset tables) with synchronous uploads; every
 * later call at that length enqueues only — dzn_segment_forward -> dzn_prepare_masks ->
 * dzn_embed_forward can then be captured in a HIP graph and replayed (r4:
 * tests/test_emb_gpu.py::test_forwards_only_enqueue_and_replay_from_a_hip_graph).
 */
#ifndef DZN_H_
#define DZN_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZN_MAX_CONV 8
#define DZN_MAX_LAYERS 32
#define DZN_MAX_HEADS 16

#define DZN_OK 0
#define DZN_E_INVALID (-1)   /* bad argument / shape / config                  */
#define DZN_E_NOMEM (-2)     /* device or host allocation failed (-> MemoryError, PA/core/inference.py:216-221) */
#define DZN_E_HIP (-3)       /* HIP runtime error                              */
#define DZN_E_STATE (-4)     /* call order violated (e.g. forward before finalize) */
#define DZN_E_MISSING (-5)   /* a required state_dict key was never loaded     */

/* element types accepted by dzn_load_tensor */
#define DZN_F32 0
#define DZN_F64 1
#define DZN_I64 2

/* arithmetic of the MFMA contractions */
#define DZN_PREC_F32 0   /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 everywhere          */
#define DZN_PREC_BF16 1  /* bf16 MFMA operands, fp32 accumulate, fp32 residual stream / norms */
#define DZN_PREC_F32_SPLIT 2 /* fp32 data everywhere; contractions split both operands exactly into
                                3 bf16 terms and run 6 bf16 MFMA products with fp32 accumulate
                                (fp32-grade accuracy, csrc/gemm_split.hip) */
#define DZN_PREC_F32_H2 3 /* fp32 data everywhere; contractions split both operands into TWO fp16 terms
                           * (exact power-of-two scaling, 22 significant bits) and sum three
                           * v_mfma_f32_16x16x32_f16 products in fp32 ("3xFP16", the error-corrected scheme
                           * known from 3xTF32): csrc/gemm_split.hip NP = 2.  Kernels without an fp16
                           * variant run as DZN_PREC_F32_SPLIT. */
#define DZN_PREC_F16 4 /* REDUCED precision (BASELINE configs[4] "fp16"): the DZN_PREC_F32_H2 engine with the
                        * linear-layer / positional-conv / ResNet contractions keeping only the LEADING fp16 term
                        * of both operands (one v_mfma_f32_16x16x32_f16 product, fp32 accumulate; the same
                        * per-window power-of-two scaling, so no fp16 range issue).  Attention, the fused
                        * conv0->conv1 frontend and the 32-channel 3x3 convolutions keep two terms; data,
                        * norms, softmax and the residual stream stay fp32.  Tolerance class: the bf16 one. */

/*
 * Architecture description.  Mirrors the kwargs of
 * diarizen/models/module/wav2vec2/model.py:779-913 (wavlm_model) as listed in
 * diarizen/models/module/wavlm_config.py, plus Model.__init__ of
 * diarizen/models/eend/model_wavlm_conformer.py:26-45.
 */
typedef struct dzn_config {
  int32_t struct_size;         /* sizeof(dzn_config), ABI check */
  int32_t precision;           /* DZN_PREC_* */
  int32_t max_batch;           /* largest B passed to *_forward */
  int32_t max_samples;         /* largest N (samples per window) */

  /* ---- WavLM frontend ---- */
  int32_t extractor_layer_norm;   /* 1: extractor_mode="layer_norm" (large), 0: "group_norm" (base) */
  int32_t normalize_waveform;     /* W2V/model.py:113 */
  int32_t n_conv;
  int32_t conv_ch[DZN_MAX_CONV];
  int32_t conv_k[DZN_MAX_CONV];
  int32_t conv_s[DZN_MAX_CONV];

  /* ---- WavLM encoder ---- */
  int32_t embed_dim;
  int32_t total_heads;
  int32_t n_layers;
  int32_t layer_norm_first;       /* encoder_layer_norm_first */
  int32_t pos_conv_kernel;
  int32_t pos_conv_groups;
  int32_t num_buckets;
  int32_t max_distance;
  int32_t use_attention[DZN_MAX_LAYERS];
  int32_t n_heads[DZN_MAX_LAYERS];                       /* len(remaining_heads[i]) */
  int32_t head_idx[DZN_MAX_LAYERS][DZN_MAX_HEADS];       /* remaining_heads[i][j]   */
  int32_t use_ffn[DZN_MAX_LAYERS];
  int32_t ffn_dim[DZN_MAX_LAYERS];

  /* ---- EEND Conformer head ---- */
  int32_t attention_in;           /* 256 */
  int32_t ffn_hidden;             /* 1024 */
  int32_t conf_heads;             /* 4 */
  int32_t conf_layers;            /* 4 */
  int32_t conf_kernel;            /* 31 */
  int32_t n_classes;              /* 11 powerset classes */
  int32_t max_speakers_per_chunk; /* 4 */
  int32_t max_speakers_per_frame; /* 2 */

  /* ---- WeSpeaker ResNet34 embedding ---- */
  int32_t has_embedding;          /* 0: segmentation only */
  int32_t embed_out_dim;          /* 256 */
  int32_t num_mel_bins;           /* 80 */
  int32_t reserved[16];
} dzn_config;

typedef struct dzn_handle dzn_handle;

/* Create an engine on the current HIP device.
 * Workspace (allocated when the weights are finalised, sized for max_batch windows of max_samples): besides the activation
 * buffers, pre-norm (layer_norm_first) encoders in the fp32 modes take n_layers further [max_batch * frames, D] fp32 buffers —
 * every layer's output rows stay in the layer's own buffer and the layer-weighted sum (model_wavlm_conformer.py:236,253-254)
 * is one pass over them after the last layer instead of a read-modify-write per layer (same additions, same order, same bits;
 * 22.6 GB at 576 windows of 8 s for wavlm-large).  They are taken only while they fit in half of the free device memory;
 * DZN_NO_WS_DEFER in the environment at that time keeps the per-layer read-modify-write. */
int dzn_create(const dzn_config* cfg, dzn_handle** out);

/*
 * Hand one state_dict entry to the engine (host memory, copied before return).
 * Keys are the reference's own:  segmentation keys as saved from Model
 * ("wavlm_model.feature_extractor.conv_layers.0.conv.weight", ..., "classifier.bias"),
 * embedding keys prefixed "embedding." + the WeSpeakerResNet34 key
 * ("embedding.resnet.conv1.weight", ...).  Unknown keys are ignored
 * (load_state_dict(strict=False) semantics) but counted; see dzn_num_ignored().
 */
int dzn_load_tensor(dzn_handle* h, const char* key, const void* host_ptr,
                    const int64_t* shape, int32_t ndim, int32_t dtype);

/* Fold weight-norm / BatchNorm / dummy_weight, pack + pad for MFMA tiles, upload,
 * allocate workspace for (max_batch, max_samples). */
int dzn_finalize_weights(dzn_handle* h);

/* number of frames L for N samples (model_wavlm_conformer.py:98-124) */
int dzn_num_frames(const dzn_handle* h, int32_t num_samples);

/*
 * Segmentation forward.  d_wave: device f32 [B, N] (channel already selected,
 * inference.py:128 / model_wavlm_conformer.py:250).  Outputs (device, either may
 * be NULL): d_logp f32 [B, L, n_classes] log-probabilities; d_multilabel u8
 * [B, L, max_speakers_per_chunk] hard multilabel decisions.
 */
int dzn_segment_forward(dzn_handle* h, const float* d_wave, int32_t B, int32_t N,
                        float* d_logp, uint8_t* d_multilabel, void* hip_stream);

/*
 * Embedding forward with the ResNet trunk shared by the S masks of a window.
 * d_wave f32 [B, N]; d_masks f32 [B, S, L]; d_emb f32 [B, S, embed_out_dim].
 */
int dzn_embed_forward(dzn_handle* h, const float* d_wave, const float* d_masks,
                      int32_t B, int32_t S, int32_t N, int32_t L, float* d_emb,
                      void* hip_stream);

/*
 * On-device glue between the two stages (keeps a window in HBM):
 *   median filter (odd `median_size`, 0/1 = off) along frames of d_multilabel u8 [B, L, S]
 *       <- diarizen/pipelines/inference.py:131-132 (scipy median_filter size=(1,11,1), "reflect")
 *   embedding masks f32 [B, S, L]: overlap excluded, fallback to the full mask when the clean
 *   mask has <= min_num_frames frames
 *       <- PA/pipelines/speaker_diarization.py:268-322
 * d_filtered u8 [B, L, S] (may be NULL) receives the filtered decisions, d_masks may be NULL.
 */
int dzn_prepare_masks(dzn_handle* h, const uint8_t* d_multilabel, int32_t B, int32_t L,
                      int32_t median_size, int32_t exclude_overlap, int32_t min_num_frames,
                      uint8_t* d_filtered, float* d_masks, void* hip_stream);

/*
 * Host post-processing on the device (stateless, no handle): the two overlap-add aggregations of
 * the pipeline over the per-window decisions that are already in HBM.
 *   dzn_speaker_count       <- SpeakerDiarizationMixin.speaker_count + Inference.aggregate(hamming=False,
 *                              skip_average=False, missing=0)   PA/pipelines/utils/diarization.py:147-155,
 *                              PA/core/inference.py:574-666:  count[t] = uint8(rint(sum_c sum_s seg / #windows))
 *   dzn_cluster_activations <- SpeakerDiarization.reconstruct + the aggregate(skip_average=True) of to_diarization
 *                              PA/pipelines/speaker_diarization.py:400-425, diarization.py:213-220:
 *                              act[t,k] = sum_c max_s{seg[c,t-start_c,s] : hard[c,s]==k}   (hard < 0 = inactive)
 * d_seg u8 [C,L,S]; d_start_frame int32 [C] = closest_frame(c*step + duration_frame/2) computed by the caller with
 * the reference's float64 arithmetic; T = number of output frames.  d_work int32 [2T] scratch; d_act int32 [T,K], K<=32.
 */
int dzn_speaker_count(const uint8_t* d_seg, int32_t C, int32_t L, int32_t S, const int32_t* d_start_frame, int32_t T,
                      int32_t* d_work, uint8_t* d_count, void* hip_stream);
int dzn_cluster_activations(const uint8_t* d_seg, const int8_t* d_hard, int32_t C, int32_t L, int32_t S,
                            const int32_t* d_start_frame, int32_t T, int32_t K, int32_t* d_act, void* hip_stream);

/* Copy a named intermediate activation of the LAST forward to host (debug / parity
 * tests).  *n_elems receives the element count; host_out may be NULL to query. */
int dzn_debug_fetch(dzn_handle* h, const char* name, float* host_out, int64_t cap,
                    int64_t* n_elems);

/* dzn_embed_forward skips the ResNet trunk for windows in which no speaker is active (their embeddings are seg_1's
 * bias: zero weights pool to zero, PA/models/blocks/pooling.py:44-131).  The subset is chosen ON THE DEVICE (r4): the
 * forward reads nothing back and stays enqueue-only.  Counters since dzn_create: windows seen by dzn_embed_forward /
 * windows whose trunk pass was skipped; they live on the device, so THIS call synchronises the device (diagnostics;
 * no reference counterpart).  DZN_EMB_NO_SKIP in the environment at dzn_create switches the subset off. */
int dzn_embed_skip_stats(const dzn_handle* h, int64_t* windows, int64_t* skipped);

int dzn_num_ignored(const dzn_handle* h);
int64_t dzn_workspace_bytes(const dzn_handle* h);
const char* dzn_last_error(const dzn_handle* h); /* h may be NULL: last create error */
int dzn_destroy(dzn_handle* h);
const char* dzn_version(void);

/*
 * Host-clustering accelerator.  Drop-in for
 *     scipy.cluster.hierarchy.linkage(emb, method="centroid", metric="euclidean")
 * as called by AgglomerativeClustering.cluster (PA/pipelines/clustering.py:407-416) and by the AHC
 * initialisation of VBxClustering (PA/pipelines/clustering.py:656-658): h_emb is a HOST float32
 * [n, dim] matrix (rows already unit-normalised by the caller, as in the reference), h_Z receives the
 * scipy dendrogram [n-1, 4] float64 (ids, ids, distance, size).  The n x n float64 distance matrix
 * lives in HBM (8 n^2 bytes -> DZN_E_NOMEM when it does not fit).  Blocking; device < 0 = current.
 */
int dzn_linkage_centroid(const float* h_emb, int32_t n, int32_t dim, double* h_Z, int32_t device);

/* Row f1, assignment step: scipy.spatial.distance.cdist(emb, centroids, metric="cosine") of
 * BaseClustering.assign_embeddings (pyannote/audio/pipelines/clustering.py:207-216) on the device: float64, row
 * norms, then 1 - clip(dot / (|u||v|)) with in-order sums and no FMA (agrees with scipy to 2e-15 on distances of
 * order 1; identical rows give identical scores).  HOST pointers: h_emb f32 [n, dim], h_cent f64 [k, dim],
 * h_dist f64 [n, k].  Blocking; device < 0 = current.  Rows with a NaN / zero norm give NaN like scipy. */
int dzn_cdist_cosine(const float* h_emb, int32_t n, int32_t dim, const double* h_cent, int32_t k, double* h_dist,
                     int32_t device);

/* (r5) dzn_linkage_centroid / dzn_cdist_cosine work on their OWN non-blocking, highest-priority stream and carve their
 * buffers from one grow-only arena per device (they are called from the pipeline's host stage, which may run in a second
 * thread while the engine executes the next recording's device stage: no null-stream ordering behind the engine's queue, no
 * hipFree).  The arena is re-allocated only when a call needs more than any call before it; these return it / report it.
 * A dzn_vbx_* state is carved from a second arena of the same context (leased to one state at a time — a second live state gets a
 * block of its own — and kept while a state holds it) and works on the same stream.
 * device < 0 in dzn_host_workspace_release = every device.  No reference counterpart (scipy owns its host buffers). */
int dzn_host_workspace_release(int32_t device);
int64_t dzn_host_workspace_bytes(int32_t device);

/* Row f1, VBx: the variational-Bayes mixture of diarizen/clustering/VBx.py:27-125 (loopProb = 0 branch :99-107, the
 * one VBxClustering.__call__ runs) with its two E-sized passes per iteration on the device (csrc/vbx.hip), float64.
 * The K x D statistics pass through the host, which keeps the reference's expressions for invL / alpha / ELBO and its
 * stopping rule (diarizen_amd/clustering.py:vb_gmm).  HOST pointers, blocking calls; device < 0 = current.
 *   dzn_vbx_create   X f64 [E, D] (PLDA-space features), Phi f64 [D], gamma0 f64 [E, K] -> opaque state
 *   dzn_vbx_stats    h_stats f64 [K, D + 1] <- (gamma^T rho | column sums of gamma), rho = X sqrt(Phi)
 *   dzn_vbx_estep    alpha f64 [K, D], ck f64 [K] = 0.5 sum_d (invL + alpha^2) Phi, lpi f64 [K] = log(pi + 1e-8):
 *                    gamma <- softmax_k(Fa (rho alpha^T - ck + G) + lpi); *h_total = sum_e logsumexp_k(...)
 *   dzn_vbx_gamma    h_gamma f64 [E, K] <- the current responsibilities
 */
int dzn_vbx_create(const double* h_X, const double* h_Phi, const double* h_gamma0, int32_t E, int32_t D, int32_t K,
                   int32_t device, void** out_state);
int dzn_vbx_stats(void* state, double* h_stats);
int dzn_vbx_estep(void* state, const double* h_alpha, const double* h_ck, const double* h_lpi, double Fa,
                  double* h_total);
int dzn_vbx_gamma(void* state, double* h_gamma);
int dzn_vbx_destroy(void* state);

/* Row f3, audio ingest: native FLAC decoder (csrc/flac.cpp, host code; the reference reads whatever torchaudio.load does,
 * diarizen/pipelines/inference.py:127 — WAV is parsed in diarizen_amd/audio.py, FLAC here).  Frame-header CRC-8 and frame
 * CRC-16 are verified while decoding; the caller checks the STREAMINFO MD5 (audio.load_flac does).
 *   dzn_flac_info    stream parameters from STREAMINFO (any output pointer may be NULL); total_samples 0 = unknown
 *   dzn_flac_decode  out = int32 [capacity_samples][channels], interleaved, sign-extended; *decoded = samples per channel.
 * Returns DZN_OK, DZN_E_INVALID (not FLAC / corrupt / unsupported), DZN_E_NOMEM (capacity too small). */
int dzn_flac_info(const uint8_t* data, size_t n, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                  int64_t* total_samples, uint8_t* md5_16);
int dzn_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t capacity_samples, int64_t* decoded);

#ifdef __cplusplus
}
#endif
#endif /* DZN_H_ */
