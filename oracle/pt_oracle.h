/*
 * pt_oracle.h -- CPU ORACLE for the ipu_path_trace hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (ipu_path_trace_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (markp-gc/ipu_path_trace) ships no tests, no golden
 * vectors and no known-answer fixtures for this path, and the third-party library that
 * holds its ray/BSDF arithmetic (external/light, https://github.com/mpups/light.git,
 * .gitmodules:1-3, pinned commit unrecoverable) is an empty, un-vendored submodule.
 * This file restates the algorithm from the reference's call sites
 * (src/codelets/codelets.cpp, src/neural_networks/NifModel.cpp, ...); every light::
 * function is marked INFERRED.  What pins it instead: analytic known-answer tests
 * (tests/test_oracle_*.py), the Random123 Philox KAT, the decode constants in the
 * reference's nif_metadata.txt and a silhouette check against images/example.png.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/codelets/TraceRecord.hpp:7-19 -- 20 bytes, offsets 0/2/4/8/12/16/18. */
typedef struct {
  uint16_t u, v;
  float r, g, b;
  uint16_t sampleCount;
  uint16_t pathLength;
} orc_trace_record;

enum { ORC_AA_NORMAL = 0, ORC_AA_UNIFORM = 1, ORC_AA_TRUNCATED_NORMAL = 2 };
enum { ORC_SAMPLES_HALF = 0, ORC_SAMPLES_FLOAT = 1 };
enum { ORC_ENV_CONSTANT = 0, ORC_ENV_NIF = 1 };
enum { ORC_FOLD_BACKWARD = 0, ORC_FOLD_FORWARD = 1 };

/* Contribution types: codelets.cpp:187,193,207,212,221 / WrappedArray.hpp:55-62. */
enum { ORC_DIFFUSE = 0, ORC_EMIT = 1, ORC_ESCAPED = 2, ORC_REFRACT = 3,
       ORC_SPECULAR = 4, ORC_DEBUG = 5, ORC_END = 6, ORC_SKIP = 7 };

typedef struct {
  uint32_t width, height;        /* PathTracerApp.cpp:799-800 */
  uint32_t max_path_length;      /* :817  (capacity of the contribution stack) */
  uint32_t roulette_depth;       /* :805 */
  float stop_prob;               /* :806  (device holds it as half: IpuPathTraceJob.cpp:137) */
  float refractive_index;        /* :804  (half on device: IpuPathTraceJob.cpp:133) */
  int32_t aa_noise_type;         /* :813 */
  int32_t sample_precision;      /* primary samples are half on the IPU (PathTracerApp.cpp:297) */
  uint64_t seed;                 /* :812 */
  float aa_noise_scale;          /* :807  (half on device: PathTracerApp.cpp:590) */
  float fov_radians;             /* :574  (half on device: PathTracerApp.cpp:591) */
  float azimuth_radians;         /* :584 */
  int32_t env_mode;
  float env_rgb[3];              /* constant sky radiance (config C1) */
  int32_t fold;                  /* ORC_FOLD_BACKWARD follows codelets.cpp:241-304 exactly */
} orc_config;

/* One dense layer: DenseLayer.hpp:18-31.  kernel is fp16 bits, row-major [rows=in][cols=out]
 * exactly as the H5 stores it (NifModel.cpp:375-401); bias fp16 [cols] or NULL. */
typedef struct {
  uint32_t rows, cols;
  const uint16_t* kernel;
  const uint16_t* bias;
  int32_t relu;                  /* activation == "relu" (NifModel.cpp:323-325) */
} orc_layer;

/* The same for a model stored as float32 (Hdf5Model.cpp:109-133 accepts float32 variables): the reference gives the
 * matmul the kernel's type (NifModel.cpp:314), so such a model runs in float -- no rounding to half between the layers. */
typedef struct {
  uint32_t rows, cols;
  const float* kernel;
  const float* bias;
  int32_t relu;
} orc_layer_f32;

/* A layer of either type, for models that mix them (float32 != 0: kernel and bias are float, else binary16 bits). */
typedef struct {
  uint32_t rows, cols;
  const void* kernel;
  const void* bias;
  int32_t relu;
  int32_t float32;
} orc_layer_any;

typedef struct orc_nif orc_nif;

typedef struct {
  uint32_t length;     /* records on the contribution stack incl. terminator (codelets.cpp:253) */
  uint32_t escaped;    /* terminal record is ESCAPED */
  float dir[3];        /* escaped ray direction (codelets.cpp:187) */
  float uv[2];         /* equirect coords (codelets.cpp:333-347) */
  float throughput[3]; /* forward product of clr*weight incl. the terminal weight */
  float cam[2];        /* half-rounded camera ray x,y (codelets.cpp:74-75) */
} orc_path;

typedef struct {
  uint64_t paths;
  uint64_t segments;   /* sum of pathLength (LoadBalancer.cpp:198-213) */
  uint64_t escaped;
} orc_stats;

/* ---- NIF (NifModel.cpp:185-245,295-326) ---- */
orc_nif* orc_nif_create(const orc_layer* layers, uint32_t n_layers, uint32_t embedding_dim,
                        float max, const float mean_folded[3], int32_t log_tonemap);
orc_nif* orc_nif_create_f32(const orc_layer_f32* layers, uint32_t n_layers, uint32_t embedding_dim,
                            float max, const float mean[3], int32_t log_tonemap);
orc_nif* orc_nif_create_mixed(const orc_layer_any* layers, uint32_t n_layers, uint32_t embedding_dim,
                              float max, const float mean[3], int32_t log_tonemap);
void orc_nif_destroy(orc_nif*);
/* u,v -> bgr (decoded).  Also the streamed-IO standalone mode of NifModel.cpp:268-278. */
int orc_nif_infer(const orc_nif*, const float* u, const float* v, size_t n, float* bgr);
/* Fourier features for one (u,v): NifModel.cpp:200-216 (fp16 trig). feats has 4*emb floats. */
void orc_nif_encode(uint32_t embedding_dim, float u, float v, float* feats);
uint64_t orc_nif_flops_per_sample(const orc_nif*);   /* NifModel.cpp:129-133 */

/* ---- path tracing (codelets.cpp) ---- */
int orc_trace_path(const orc_config*, uint16_t u, uint16_t v, uint32_t sample_index, orc_path*);
/* Dump the contribution stack of one path: types[i], clr[3*i..], weight[i]; returns count. */
int orc_trace_records(const orc_config*, uint16_t u, uint16_t v, uint32_t sample_index,
                      int32_t* types, float* clr, float* weight, uint32_t capacity);
/* K2..K12 for samples [sample_base, sample_base+n_samples) over a worklist; accumulates
 * into records exactly as AccumulateContributions (codelets.cpp:294-300). */
int orc_render(const orc_config*, const orc_nif*, orc_trace_record* records, size_t n,
               uint32_t sample_base, uint32_t n_samples, orc_stats* stats);
int orc_max_threads(void);
const char* orc_build_info(void);   /* "strict ..." or the timing build's flags (libpt_oracle_fast.so) */

/* ---- building blocks exposed for known-answer tests ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint16_t orc_f2h(float);
float orc_h2f(uint16_t);
float orc_dm_log(float);
void orc_dm_sincos2pi(float u, float* s, float* c);
float orc_dm_atan2(float y, float x);
float orc_dm_acos(float x);
void orc_pixel_to_ray(float col, float row, uint32_t w, uint32_t h, float fov, float out[3]);
float orc_intersect_sphere(const float o[3], const float d[3], const float c[3], float radius);
float orc_intersect_disc(const float o[3], const float d[3], const float n[3], const float c[3], float radius);
/* returns object index 0..5 or -1; t, hit point and normal written when hit. */
int orc_scene_intersect(const float o[3], const float d[3], float* t, float hp[3], float nrm[3]);
void orc_reflect(float d[3], const float n[3]);
int orc_refract(float d[3], const float n[3], float ri, float u);
void orc_hemisphere(float u1, float u2, float out[3]);
void orc_diffuse_dir(const float n[3], float u1, float u2, float out[3]);
int orc_roulette(float u, float stop_prob, float* factor);
void orc_dir_to_uv(const float d[3], float azimuth, float uv[2]);
void orc_aa_noise(const orc_config*, uint16_t u, uint16_t v, uint32_t sample_index, float noise[2]);
void orc_scene_object(int index, float centre[3], float* radius, float colour[3], int32_t* type);
/* object index per pixel (central ray), -1 = environment */
void orc_object_ids(uint32_t w, uint32_t h, float fov, int8_t* ids);
void orc_specular_ids(uint32_t w, uint32_t h, float fov, float ri, int reflect_variant, int max_bounces, int8_t* ids,
                      uint8_t* bounces);

#ifdef __cplusplus
}
#endif
#endif
