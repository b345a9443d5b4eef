/*
 * libstarvector_hip.so - C ABI of the MI355X (gfx950) StarVector im2svg inference engine.
 *
 * Drop-in boundary for the hot path of joanrod/star-vector (paths below are relative to the
 * reference checkout):
 *
 *   sv_encode_image        replaces  ImageEncoder.forward, clip branch
 *                                    starvector/model/image_encoder/image_encoder.py:91-94
 *                                    (VisionTransformer.forward, .../clip_model.py:181-191)
 *   sv_adapter             replaces  Adapter.forward  starvector/model/adapters/adapter.py:33-39
 *   sv_embed_tokens        replaces  StarVectorStarCoder._get_embeddings (wte lookup)
 *                                    starvector/model/models/starvector_v1.py:16-18
 *   sv_prefill / sv_decode_step
 *                          replace   GPTBigCodeForCausalLM.forward as called by HF generate
 *                                    (restated in starvector/model/gpt_bigcode/modeling_gpt_bigcode.py:930-1134,1241-1258)
 *   sv_generate            replaces  svg_transformer.transformer.generate(**generation_kwargs)
 *                                    starvector/model/models/starvector_base.py:228-241,255
 *                                    incl. StoppingCriteriaSub (starvector_base.py:9-20)
 *   sv_load_weight         ingests the reference state_dict keys (train/util.py:71 naming;
 *                          SURVEY.md section 8b "Weight names")
 *
 * This header is the PRODUCT ABI: what a reference-side binding needs -- engine life cycle, weights,
 * the forward entry points, generate, continuous batching, the beam scorer, image pre-processing.  The test / measurement surface
 * (sv_op_* single operators, sv_debug_* plans and traces, sv_bench_*, sv_profile_*) is compiled into the same library but declared
 * in include/starvector_hip_debug.h: nothing there is needed to run the path, and three of its switches change which kernels a live
 * engine launches (A/B measurements only).
 *
 * Conventions: every pointer named dev_* / documented "device" is a HIP device pointer owned by the
 * caller (PyTorch-ROCm tensor.data_ptr()); `stream` is a hipStream_t passed as void* (0 = default
 * stream); all work is enqueued on it.  Functions return 0 on success or a negative SV_E* code;
 * sv_last_error() returns the message of the calling thread's last failure.  One in-flight call
 * per engine handle (internal mutex); handles are independent.  bf16 = IEEE bfloat16 bits.
 */
#ifndef STARVECTOR_HIP_H
#define STARVECTOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_ABI_VERSION 8
#define SV_WEIGHT_BF16 0
#define SV_WEIGHT_FP8_E4M3 1

enum {
    SV_OK = 0,
    SV_EINVAL = -22,      /* bad argument / shape */
    SV_ENOMEM = -12,      /* device allocation failed */
    SV_ENOENT = -2,       /* unknown weight name / missing weight */
    SV_EHIP = -5,         /* HIP runtime error */
    SV_ESTATE = -1,       /* call order (e.g. generate before weights are complete) */
    SV_ENOTSUP = -95,     /* a combination the engine does not implement (says which) */
    SV_EBUSY = -16        /* continuous batching: no free slot / KV pages for the request right now (release finished slots, retry) */
};

enum { SV_DTYPE_BF16 = 0, SV_DTYPE_F32 = 1 };
enum { SV_ARCH_V1 = 0, SV_ARCH_V2 = 1 };
enum { SV_NORM_LAYER = 0, SV_NORM_BATCH = 1 };          /* adapter_norm (adapter.py:25-28) */
enum { SV_ACT_NONE = 0, SV_ACT_QUICKGELU = 1, SV_ACT_SWISH = 2, SV_ACT_GELU_TANH = 3 };

typedef struct sv_engine sv_engine;
typedef void* sv_stream;

/* Shapes of StarVector-1B unless overridden (StarVectorConfig, starvector_arch.py:96-131) */
typedef struct sv_config {
    int32_t image_size;        /* 224 */
    int32_t patch_size;        /* 14  */
    int32_t vit_width;         /* 1024 */
    int32_t vit_layers;        /* 23 (image_encoder.py:52-58) */
    int32_t vit_heads;         /* 16 */
    int32_t adapter_norm;      /* SV_NORM_LAYER */
    int32_t hidden;            /* 2048 */
    int32_t n_layer;           /* 24 */
    int32_t n_head;            /* 16 */
    int32_t n_inner;           /* 8192 */
    int32_t vocab;             /* 49156 */
    int32_t n_positions;       /* 8192 */
    int32_t max_batch;         /* sequences per generate call (KV pool + workspace sizing) */
    int32_t max_seq_len;       /* prompt + generated tokens per sequence (<= n_positions) */
    float   ln_eps;            /* 1e-5 */
    int32_t device;            /* HIP device ordinal */
    /* StarVector-8B; sv_config_default_1b() zeroes / defaults them */
    int32_t arch;              /* SV_ARCH_V1: CLIP + GPTBigCode (MQA, learned positions)
                                  SV_ARCH_V2: SigLIP tower + StarCoder2 (RoPE, GQA)   */
    int32_t n_kv_head;         /* v2: key/value heads (v1: 1) */
    float   rope_theta;        /* v2: rotary base (1e6 for bigcode/starcoder2-7b) */
    int32_t vit_mlp;           /* v2: SigLIP intermediate size (v1: 4 * vit_width) */
    float   vit_eps;           /* v2: SigLIP layer_norm_eps 1e-6 (v1: ln_eps) */
    int32_t sliding_window;    /* v2: a query sees the last `sliding_window` keys, itself included (4096 for
                                  bigcode/starcoder2-7b; HF's eager/sdpa mask `kv > q - W`); 0 = full attention.
                                  The prompt must fit inside the window. */
    int32_t weight_dtype;      /* SV_WEIGHT_BF16 (reference precision) or SV_WEIGHT_FP8_E4M3: decoder Linear weights and the
                                  lm_head are quantised at load to OCP e4m3 with one scale per output row (BASELINE config 5,
                                  "fp8 weights"): the decode step streams half the bytes; activations, accumulation and every
                                  other tensor stay as they are.  Not a reference numerics mode: see DESIGN.md. */
    int32_t exclusive_device;  /* != 0: this engine is the only thing launching kernels on its GPU while it decodes (the deployment the
                                  path is built for: one process per GPU).  Enables the decode launches that need ALL their
                                  workgroups resident at once (mlp_fused_kernel: the MLP half of a layer as one launch, 256 blocks, one
                                  per CU, in-launch hand-off; rowln_cattn_kernel: the row update and the c_attn projection as one
                                  launch).  Results are bit-identical either way.  With ANOTHER process or engine decoding on the same
                                  GPU such a launch can find its blocks only partly resident: the waiting blocks then give up after
                                  a wall-clock bound (5 ms) and the call -- sv_generate, beam search, sv_decode_step, sv_cb_step --
                                  fails with SV_EHIP and a message naming the launch (never a hang, never tokens; executed by
                                  tests/test_gpu_safety.py); the next call starts clean.  Leave it 0 there.  Default 0.
                                  2 = OPTIMISTIC: as 1 until a fused launch gives up for the first time; the engine then switches those launches
                                  off for the rest of its life (one line on stderr) and sv_generate runs the failed call again without them -- the
                                  kernels are bit-identical either way, so the caller sees the same tokens, late by the failed attempt (<= 80 ms
                                  in the safety test), and a streaming callback gets every column exactly once; sv_decode_step / sv_cb_step report
                                  that one failure and are clean from the next call on.  What the Python wrapper passes by default. */
} sv_config;

/* Streaming: called on the host with the tokens that became final since the last call -- tokens [batch][n_cols] int32
 * row-major, covering output columns first_col .. first_col + n_cols - 1 -- every `sync_every` steps and once at the end
 * (what a HF `streamer` gets through put(); the reference's serve worker builds one, serve/model_worker.py:129,170, but its
 * kwargs whitelist never lets it reach generate). */
typedef void (*sv_token_callback)(void* user, const int32_t* tokens, int32_t batch, int32_t first_col, int32_t n_cols);

/* generate(...) arguments that reach HF generate through starvector_base.py:228-241 */
typedef struct sv_sampling {
    int32_t do_sample;         /* 0 = greedy argmax */
    float   temperature;       /* used when do_sample */
    float   top_p;             /* used when do_sample */
    int32_t max_length;        /* HF semantics: INCLUDES the prompt rows (budget = max_length - S0) */
    int32_t eos_token_id;
    int32_t pad_token_id;
    int32_t n_stop;            /* length of the row-0 stop sequence ("</svg>" ids), 0 = none */
    const int32_t* stop_ids;   /* host pointer, n_stop entries */
    uint64_t seed;             /* sampling RNG seed */
    int32_t sync_every;        /* host polls the device "done" flag every this many steps (0 = 32) */
    float   repetition_penalty;/* HF RepetitionPenaltyLogitsProcessor over the generated ids; 0 or 1 = off */
    int32_t num_beams;         /* 0 or 1 = greedy / sampling; 2..8 = HF beam search (the reference's default is 2,
                                  starvector_base.py:234); batch * num_beams <= max_batch; with do_sample: beam-sample */
    float   length_penalty;    /* beam search: hypothesis score = sum log-probs / len ** length_penalty (:238) */
    int32_t early_stopping;    /* beam search: 0 False (HF default), 1 True (:293), 2 "never" */
    int32_t top_k;             /* used when do_sample: HF TopKLogitsWarper before top-p; 0 = off.  The reference never
                                  passes it, but its pinned transformers==4.49.0 defaults GenerationConfig.top_k to 50 */
    sv_token_callback on_tokens; /* optional streaming callback (NULL = off); not with num_beams > 1 (as in HF) */
    void*   user_data;
    int32_t min_new_tokens;    /* HF MinLengthLogitsProcessor: the EOS logit is held at -inf while fewer than this many tokens
                                  have been generated.  The caller passes max(min_length - S0, 0) (HF subtracts the prompt
                                  length when generating from inputs_embeds; starvector_base.py:236 passes min_length).
                                  0 = off.  With num_beams > 1 HF applies it to the log-probabilities: so does the scorer */
} sv_sampling;

/* HF beam search bookkeeping as a standalone device-side scorer (what transformers' _beam_search does between two
 * forward passes): feed it the [batch * num_beams][vocab] fp32 logits of each step.  sv_generate drives the same
 * object internally; this handle exists so the parity tests can check it against the oracle on synthetic logits. */
typedef struct sv_beam sv_beam;
typedef struct sv_beam_config {
    int32_t batch, num_beams, vocab, max_new;
    int32_t eos_token_id, pad_token_id;
    int32_t early_stopping;    /* 0 False, 1 True, 2 "never" */
    float   length_penalty;
    float   repetition_penalty;
    int32_t n_stop;            /* the reference's row-0 stop sequence (starvector_base.py:9-20) */
    const int32_t* stop_ids;   /* host pointer */
    int32_t do_sample;         /* beam-sample: warpers (temperature, top_k, top_p; min_tokens_to_keep 2) on the log-probs,
                                  then 2*num_beams draws without replacement from softmax(accumulated scores) */
    float   temperature, top_p;
    int32_t top_k;
    uint64_t seed;
    int32_t min_new_tokens;    /* HF MinLengthLogitsProcessor on the log-probabilities (as _beam_search applies processors) */
} sv_beam_config;

int  sv_abi_version(void);
const char* sv_last_error(void);
void sv_config_default_1b(sv_config* cfg);
void sv_config_default_8b(sv_config* cfg);    /* siglip_384 + starcoder2-7b shapes, max_batch 16 */

int  sv_create(const sv_config* cfg, sv_engine** out);
int  sv_destroy(sv_engine* e);

/* Ingest one tensor of the reference state_dict (device pointer, dtype SV_DTYPE_*).  Linear weights
 * are repacked into MFMA fragment order in library-owned memory; the caller's tensor is only read. */
int  sv_load_weight(sv_engine* e, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim,
                    const int64_t* shape, sv_stream stream);
/* 0 when every tensor the path needs has been loaded; otherwise SV_ENOENT and sv_last_error() names
 * the first missing key. */
int  sv_weights_complete(sv_engine* e);
/* The tensors an engine of this configuration expects, sorted by name: reference state_dict key (train/util.py:71 names), element
 * count, whether sv_weights_complete insists on it (the tied lm_head is optional) and whether it has been loaded.
 * sv_weight_count returns the number of entries (>= 0) or a negative error code. */
int  sv_weight_count(sv_engine* e);
int  sv_weight_info(sv_engine* e, int32_t index, char* name, int32_t name_cap, int64_t* numel, int32_t* required, int32_t* loaded);

/* image [B,3,S,S] bf16 (device) -> out [B, T, vit_width] bf16, T = (S/patch)^2 + 1 */
int  sv_encode_image(sv_engine* e, const void* dev_image, int32_t B, void* dev_out, sv_stream stream);
/* in [B,T,vit_width] bf16 -> out [B,T,hidden] bf16 */
int  sv_adapter(sv_engine* e, const void* dev_in, int32_t B, void* dev_out, sv_stream stream);
/* ids [n] int64 (device) -> out [n, hidden] bf16 */
int  sv_embed_tokens(sv_engine* e, const int64_t* dev_ids, int32_t n, void* dev_out, sv_stream stream);
/* The same two ops writing STRAIGHT into the inputs_embeds buffer [B][S0][hidden] that sv_prefill / sv_generate read (starvector_base.py:
 * 203-221 without the torch.cat): image b's visual rows -> rows 0 .. T-1 of its block; the P token rows of sequence b -> rows
 * row0 .. row0 + P - 1 (row0 = T for im2svg).  dev_ids: int64 [B][P]. */
int  sv_adapter_into(sv_engine* e, const void* dev_in, int32_t B, void* dev_embeds, int32_t S0, sv_stream stream);
int  sv_embed_tokens_into(sv_engine* e, const int64_t* dev_ids, int32_t B, int32_t P, void* dev_embeds, int32_t S0, int32_t row0,
                          sv_stream stream);

/* Image pre-processing on device = `ImageTrainProcessor.__call__` (starvector/data/util.py:40-68):
 * dev_pixels uint8 [height][width][channels] (3 = RGB, 4 = RGBA composited on white) -> white pad to square -> Pillow's
 * antialiased BICUBIC resize to out_size -> /255 -> (x - mean) / std.  dev_out float32 [3][out_size][out_size], bit for bit
 * the tensor the reference computes with Pillow + torchvision.  Needs no engine handle.
 * recipe 0: the above (clip branch).  recipe 1: the SigLIP tower's HF image processor (image_encoder.py:45-48,116-117):
 * alpha dropped (`convert("RGB")`), the image STRETCHED to out_size x out_size by the same resampler, rescale by 1/255 in
 * double, then (x - mean) / std (mean = std = 0.5 for google/siglip-large-patch16-384). */
int  sv_preprocess_image(const uint8_t* dev_pixels, int32_t width, int32_t height, int32_t channels, int32_t out_size,
                         int32_t recipe, const float* mean3, const float* std3, float* dev_out, sv_stream stream);

/* The same recipe for a BATCH of images of any sizes (the serving TTFT path: one call, three launches per 32 images):
 * dev_pixels / widths / heights / channels are HOST arrays of n entries (device pointers in dev_pixels); dev_out is float32
 * [n][3][out_size][out_size].  Stateless: the caller owns the workspace (sv_preprocess_workspace_bytes, a host computation),
 * the fixed-point tap tables are computed on device into it, every per-image parameter travels in the kernel arguments --
 * no host copy, no lock, no stream synchronisation, no library-owned buffer. */
int64_t sv_preprocess_workspace_bytes(const int32_t* widths, const int32_t* heights, int32_t n, int32_t out_size, int32_t recipe);
int  sv_preprocess_images(const uint8_t* const* dev_pixels, const int32_t* widths, const int32_t* heights,
                          const int32_t* channels, int32_t n, int32_t out_size, int32_t recipe, const float* mean3,
                          const float* std3, float* dev_out, void* dev_workspace, int64_t workspace_bytes, sv_stream stream);

/* Prompt pass over inputs_embeds [B,S0,hidden] bf16 (all-ones attention mask): fills the paged KV
 * cache and writes the last-row logits [B, vocab] fp32 (bf16-rounded values, as the reference's
 * bf16 lm_head produces). */
int  sv_prefill(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, float* dev_logits,
                sv_stream stream);
/* Scoring forward = `StarVectorForCausalLM.forward(vision_embeds, input_ids, ..., num_logits_to_keep)`
 * (starvector_arch.py:161-184; GRPO's log-prob pass, inference mode): the decoder over inputs_embeds [B,S,hidden] bf16
 * (all-ones mask) and the lm_head over the LAST n_keep positions of every sequence.
 * dev_logits_bf16: [B, n_keep, vocab] bf16, the dtype the reference's bf16 lm_head returns.  Also leaves the KV cache
 * filled like sv_prefill. */
int  sv_forward_logits(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S, int32_t n_keep,
                       void* dev_logits_bf16, sv_stream stream);

/* One autoregressive step for tokens [B] int32 (device) appended after the cached context. */
int  sv_decode_step(sv_engine* e, const int32_t* dev_tokens, int32_t B, float* dev_logits, sv_stream stream);

/* Full generation.  out_tokens [B, max_new] int64 (device, max_new = max_length - S0) receives the
 * NEW tokens only (HF semantics); *n_generated = number of columns produced (<= max_new); trailing
 * columns are left untouched.  Blocks until generation has finished (polls a device flag). */
int  sv_generate(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, const sv_sampling* sp,
                 int64_t* dev_out_tokens, int32_t* n_generated, sv_stream stream);
/* ---- continuous batching (the reference worker's 5 concurrent requests, serve/model_worker.py:161-172,
 * 216-229, as ONE decode loop).  Every row ("slot") of the engine's batch is an independent request: own sampling parameters,
 * budget, EOS, stop sequence (the reference's row-0 stop, starvector_base.py:9-20, is right for one request per generate call
 * and becomes each request's own stop here) and random stream.  A request yields the same tokens as when it runs alone through
 * sv_generate.  While a continuous batch is active the classic entry points (sv_prefill / sv_generate / ...) return SV_ESTATE.
 *   sv_cb_admit    prompt pass of n new requests of equal prompt length S0 (dev_embeds [n][S0][hidden] bf16) into free slots --
 *                  the live slots keep their KV pages -- and their first token; slots_out [n] = the slots taken.
 *                  SV_EBUSY when slots or KV pages are short (nothing is admitted then).
 *   sv_cb_step     n_steps decode steps for all live slots (one captured hipGraph per row bucket, kept on the engine);
 *                  *n_live = slots still generating afterwards.  Finished slots idle until released.
 *   sv_cb_poll     per slot: live flag and number of tokens emitted so far (host arrays of `capacity` >= max_batch entries)
 *   sv_cb_read     tokens [first, first + count) of a slot -> host int64
 *   sv_cb_release  frees a slot and its KV pages (stops it if it is still generating)
 *   sv_cb_reset    releases everything; the engine is back to the classic entry points */
typedef struct sv_cb_request {
    int32_t do_sample; float temperature; float top_p; int32_t top_k;     /* as in sv_sampling */
    uint64_t seed;
    int32_t max_new_tokens;    /* new-token budget of THIS request (HF: max_length - prompt length) */
    int32_t eos_token_id, pad_token_id;
    int32_t min_new_tokens;
    float   repetition_penalty;
    int32_t n_stop;            /* 0..16 */
    int32_t stop_ids[16];
} sv_cb_request;
int  sv_cb_admit(sv_engine* e, const void* dev_embeds, int32_t n, int32_t S0, const sv_cb_request* reqs, int32_t* slots_out,
                 sv_stream stream);
int  sv_cb_step(sv_engine* e, int32_t n_steps, int32_t* n_live, sv_stream stream);
int  sv_cb_poll(sv_engine* e, int32_t* host_live, int32_t* host_steps, int32_t capacity);
int  sv_cb_read(sv_engine* e, int32_t slot, int32_t first, int32_t count, int64_t* host_tokens);
int  sv_cb_release(sv_engine* e, int32_t slot);
int  sv_cb_reset(sv_engine* e);

/* beam scorer: one step consumes dev_logits [batch * num_beams][ld] (row r = request r / num_beams, beam r % num_beams)
 * and reports (host arrays, each batch * num_beams long, may be NULL) the flat parent row, the token and the running
 * score of every beam that continues; *done = 1 once HF's loop would stop.  sv_beam_finalize returns the best
 * hypothesis per request: host_tokens [batch][max_new] (filled with pad-or-eos), *n_generated = common length. */
int  sv_beam_create(const sv_beam_config* cfg, sv_beam** out);
int  sv_beam_destroy(sv_beam* b);
int  sv_beam_step(sv_beam* b, const float* dev_logits, int32_t ld, int32_t* done, int32_t* host_parent,
                  int32_t* host_tokens, float* host_scores, sv_stream stream);
int  sv_beam_finalize(sv_beam* b, int64_t* host_tokens, int32_t* n_generated, float* host_scores, sv_stream stream);

/* Search trace of the last beam-search sv_generate on this engine (parity tests replay it through the oracle):
 * host_parent / host_tok [n_steps][rows], rows = batch * num_beams; entry (t, r) = the beam (index inside its request)
 * that running beam r descended from at step t, and the token it took.  NULL arrays: only *n_steps / *rows. */
int  sv_beam_history(sv_engine* e, int32_t* host_parent, int32_t* host_tok, int32_t capacity_steps, int32_t* n_steps,
                     int32_t* rows);

/* host wall-clock of the last sv_generate, 4 doubles: [0] ms prefill + first token (TTFT),
 * [1] ms decode loop, [2] decode steps enqueued, [3] decode steps per hipGraph launch (0: plain launches, 1: one step per replay,
 * > 1: the step captured that many times into one graph for calls long enough to pay for it -- SV_GRAPH_STEPS, default 32, 1 = off) */
int  sv_last_timing(sv_engine* e, double* out4);

#ifdef __cplusplus
}
#endif
#endif /* STARVECTOR_HIP_H */
