// Module-level entry points: a MODEL IMAGE (marigold_amd/image.py::export_model_image - the pipeline's three native programs
// for one problem shape, their kernel-ready weights and a memory plan) loaded and run from C, no Python at run time.
// These are the seams the reference's single_infer calls (marigold/marigold_depth_pipeline.py:396-477):
//   mg_model_vae_encode  = encode_rgb            (:479-496: vae.encoder -> quant_conv -> mean -> x 0.18215)
//   mg_model_denoise     = the T-step loop       (:455-468: unet(cat(rgb_latent, x), t, ctx) + scheduler.step, all steps)
//   mg_model_vae_decode  = decode_depth / _normals (:498-516 + :473-475: post_quant_conv -> decoder -> channel mean / clip / shift)
// File layout (little endian; written by image.py with struct.pack - the two sides are kept in step by tests/test_host.py):
//   header "MGIMG1", version, ABI, counts, cfg[16] | buffer table | program table (+ named slots) | blobs (64-byte aligned)
#include <stdio.h>
#include <string.h>

#include <exception>
#include <string>
#include <vector>

#include "common.h"

extern "C" int mg_ens_align_minimize(const mg_op* reg_op, void* stream, int E, int affine, int reduction, double lam, const double* mean,
                                     const double* C, float* st_host, const float* mm_host, double* x, double gtol, int maxiter,
                                     double* fval, int* nit, int* nfev, int* status);

namespace {

#pragma pack(push, 1)
struct ImgHeader {
  char magic[8];
  uint32_t version, abi, n_buffers, n_programs;
  uint32_t cfg[16];   // B, H, W, h, w, steps, prediction channels, post, step noises, sizeof(mg_op), modalities, Hout, Wout
};
struct ImgBuffer {
  uint64_t nbytes, file_off;
  uint32_t kind, pad;   // 0 scratch, 1 zeroed state, 2 data (weights / constants)
};
struct ImgSlot {
  char name[24];
  uint32_t buf, pad;
  uint64_t off, nbytes;
};
struct ImgProgram {
  char name[32];
  uint32_t n_ops, n_relocs;
  uint64_t ops_off, relocs_off;
  uint32_t n_slots, pad;
  ImgSlot slots[16];
};
struct ImgReloc {
  uint32_t op, slot, buf, pad;
  uint64_t off;
};
#pragma pack(pop)

struct Slot {
  std::string name;
  char* ptr;
  uint64_t nbytes;
};
struct Prog {
  std::string name;
  mg_program* prog = nullptr;
  std::vector<Slot> slots;
  const Slot* slot(const char* n) const {
    for (const Slot& s : slots)
      if (s.name == n) return &s;
    return nullptr;
  }
};

}  // namespace

struct mg_model {
  ImgHeader hdr;
  bool host_only = false;
  char* arena = nullptr;       // one device allocation holding every buffer (host-only: a fake base address, never touched)
  uint64_t arena_bytes = 0;
  std::vector<char*> bufs;
  std::vector<uint64_t> buf_bytes;   // the buffer table's sizes: every relocation / slot is checked against them
  Prog enc, den, dec;
};

namespace {

bool read_at(FILE* f, uint64_t off, void* dst, size_t n) {
  return fseek(f, (long)off, SEEK_SET) == 0 && fread(dst, 1, n, f) == n;
}

int load_program(FILE* f, const ImgProgram& ip, mg_model* m, Prog* out) {
  out->name = std::string(ip.name, strnlen(ip.name, sizeof(ip.name)));
  MG_REQUIRE(ip.n_ops > 0 && ip.n_slots <= 16, "mg_model_load: corrupt program table (%s)", out->name.c_str());
  std::vector<mg_op> ops(ip.n_ops);
  MG_REQUIRE(read_at(f, ip.ops_off, ops.data(), sizeof(mg_op) * ip.n_ops), "mg_model_load: short read (ops of %s)", out->name.c_str());
  std::vector<ImgReloc> rel(ip.n_relocs);
  MG_REQUIRE(ip.n_relocs == 0 || read_at(f, ip.relocs_off, rel.data(), sizeof(ImgReloc) * ip.n_relocs),
             "mg_model_load: short read (relocations of %s)", out->name.c_str());
  for (const ImgReloc& r : rel) {
    MG_REQUIRE(r.op < ip.n_ops && r.buf < m->bufs.size(), "mg_model_load: relocation out of range (%s)", out->name.c_str());
    MG_REQUIRE(r.off < m->buf_bytes[r.buf], "mg_model_load: corrupt image - a relocation of %s points %llu bytes into buffer %u of %llu bytes",
               out->name.c_str(), (unsigned long long)r.off, r.buf, (unsigned long long)m->buf_bytes[r.buf]);
    char* p = m->bufs[r.buf] + r.off;
    if (r.slot < 16) {
      ops[r.op].p[r.slot] = p;
    } else {   // the (i[29], i[30]) address pair of MG_OP_IGEMM's row-statistics tickets
      MG_REQUIRE(r.slot == 100 && ops[r.op].kind == MG_OP_IGEMM, "mg_model_load: unknown relocation slot %u", r.slot);
      const uint64_t a = (uint64_t)(uintptr_t)p;
      ops[r.op].i[29] = (int32_t)(uint32_t)(a & 0xffffffffu);
      ops[r.op].i[30] = (int32_t)(uint32_t)(a >> 32);
    }
  }
  out->prog = mg_program_create(ops.data(), (int)ops.size());
  MG_REQUIRE(out->prog, "mg_model_load: mg_program_create failed (%s)", out->name.c_str());
  for (uint32_t k = 0; k < ip.n_slots; ++k) {
    const ImgSlot& s = ip.slots[k];
    MG_REQUIRE(s.buf < m->bufs.size(), "mg_model_load: slot out of range (%s)", out->name.c_str());
    MG_REQUIRE(s.off <= m->buf_bytes[s.buf] && s.nbytes <= m->buf_bytes[s.buf] - s.off,
               "mg_model_load: corrupt image - slot %u of %s spans [%llu, +%llu) of a %llu-byte buffer", k, out->name.c_str(),
               (unsigned long long)s.off, (unsigned long long)s.nbytes, (unsigned long long)m->buf_bytes[s.buf]);
    out->slots.push_back(Slot{std::string(s.name, strnlen(s.name, sizeof(s.name))), m->bufs[s.buf] + s.off, s.nbytes});
  }
  return 0;
}

int load_into(mg_model* m, FILE* f, int device) {
  MG_REQUIRE(read_at(f, 0, &m->hdr, sizeof(ImgHeader)), "mg_model_load: short read (header)");
  const ImgHeader& h = m->hdr;
  MG_REQUIRE(memcmp(h.magic, "MGIMG1\0\0", 8) == 0 && h.version == 1, "mg_model_load: not a model image (or an unknown version)");
  MG_REQUIRE(h.abi == MG_ABI_VERSION && h.cfg[9] == sizeof(mg_op),
             "mg_model_load: the image was written for ABI %u / %u-byte ops, this library is ABI %d / %zu", h.abi, h.cfg[9], MG_ABI_VERSION, sizeof(mg_op));
  MG_REQUIRE(h.n_programs == 3 && h.n_buffers > 0 && h.n_buffers < (1u << 24), "mg_model_load: corrupt header");
  std::vector<ImgBuffer> bt(h.n_buffers);
  MG_REQUIRE(read_at(f, sizeof(ImgHeader), bt.data(), sizeof(ImgBuffer) * h.n_buffers), "mg_model_load: short read (buffer table)");
  // the file's own size bounds every stored buffer (a truncated / corrupt table must not size a staging buffer or a copy)
  MG_REQUIRE(fseek(f, 0, SEEK_END) == 0, "mg_model_load: cannot seek");
  const long fsz = ftell(f);
  MG_REQUIRE(fsz > 0, "mg_model_load: cannot size the image");
  const uint64_t file_bytes = (uint64_t)fsz;
  std::vector<uint64_t> off(h.n_buffers);
  uint64_t total = 0;
  m->buf_bytes.resize(h.n_buffers);
  for (uint32_t i = 0; i < h.n_buffers; ++i) {
    MG_REQUIRE(bt[i].nbytes > 0 && bt[i].nbytes < (1ull << 40), "mg_model_load: corrupt image - buffer %u has %llu bytes", i, (unsigned long long)bt[i].nbytes);
    if (bt[i].kind == 2)
      MG_REQUIRE(bt[i].file_off <= file_bytes && bt[i].nbytes <= file_bytes - bt[i].file_off,
                 "mg_model_load: corrupt image - buffer %u lies at [%llu, +%llu) of a %llu-byte file", i, (unsigned long long)bt[i].file_off,
                 (unsigned long long)bt[i].nbytes, (unsigned long long)file_bytes);
    m->buf_bytes[i] = bt[i].nbytes;
    off[i] = total;
    total += (bt[i].nbytes + 255) / 256 * 256;
  }
  m->arena_bytes = total;
  m->host_only = device < 0;
  if (m->host_only) {
    m->arena = (char*)(uintptr_t)0x100000000ull;   // addresses for the contract checks only: nothing is dereferenced
  } else {
    MG_REQUIRE(mg_init(device) == 0, "mg_model_load: %s", mg_last_error());
    MG_CHECK_HIP(hipMalloc((void**)&m->arena, total));
  }
  m->bufs.resize(h.n_buffers);
  std::vector<char> stage;
  for (uint32_t i = 0; i < h.n_buffers; ++i) {
    m->bufs[i] = m->arena + off[i];
    if (m->host_only) continue;
    if (bt[i].kind == 2) {
      stage.resize(bt[i].nbytes);
      MG_REQUIRE(read_at(f, bt[i].file_off, stage.data(), bt[i].nbytes), "mg_model_load: short read (buffer %u)", i);
      MG_CHECK_HIP(hipMemcpy(m->bufs[i], stage.data(), bt[i].nbytes, hipMemcpyHostToDevice));
    } else if (bt[i].kind == 1) {
      MG_CHECK_HIP(hipMemset(m->bufs[i], 0, bt[i].nbytes));
    }
  }
  std::vector<ImgProgram> pt(h.n_programs);
  MG_REQUIRE(read_at(f, sizeof(ImgHeader) + sizeof(ImgBuffer) * h.n_buffers, pt.data(), sizeof(ImgProgram) * h.n_programs),
             "mg_model_load: short read (program table)");
  Prog* dst[3] = {&m->enc, &m->den, &m->dec};
  const char* want[3] = {"vae.encode", "denoise", "vae.decode"};
  for (int k = 0; k < 3; ++k) {
    if (int rc = load_program(f, pt[k], m, dst[k])) return rc;
    MG_REQUIRE(dst[k]->name == want[k], "mg_model_load: program %d is '%s', expected '%s'", k, dst[k]->name.c_str(), want[k]);
  }
  MG_REQUIRE(m->enc.slot("rgb") && m->enc.slot("latent") && m->den.slot("rgb_latent") && m->den.slot("x") && m->dec.slot("latent") &&
             m->dec.slot("pred"), "mg_model_load: a program lacks its input / output slots");
  return 0;
}

int copy_dd(void* dst, const void* src, uint64_t n, hipStream_t s) {
  MG_CHECK_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
  return 0;
}

}  // namespace

extern "C" {

mg_model* mg_model_load(const char* path, int device) {
  if (!path) {
    mg_set_error("mg_model_load: null path");
    return nullptr;
  }
  FILE* f = fopen(path, "rb");
  if (!f) {
    mg_set_error("mg_model_load: cannot open %s", path);
    return nullptr;
  }
  mg_model* m = new mg_model();
  int rc;
  try {   // (std::vector / std::string allocations: nothing may unwind across the C boundary)
    rc = load_into(m, f, device);
  } catch (const std::exception& e) {
    mg_set_error("mg_model_load: %s while reading %s (corrupt image?)", e.what(), path);
    rc = 2;
  }
  fclose(f);
  if (rc) {
    mg_model_destroy(m);
    return nullptr;
  }
  return m;
}

void mg_model_destroy(mg_model* m) {
  if (!m) return;
  for (Prog* p : {&m->enc, &m->den, &m->dec})
    if (p->prog) mg_program_destroy(p->prog);
  if (m->arena && !m->host_only) (void)hipFree(m->arena);
  delete m;
}

int mg_model_info(const mg_model* m, int* cfg16) {
  MG_REQUIRE(m && cfg16, "mg_model_info: null argument");
  for (int i = 0; i < 16; ++i) cfg16[i] = (int)m->hdr.cfg[i];
  return 0;
}

long long mg_model_device_bytes(const mg_model* m) { return m ? (long long)m->arena_bytes : 0; }

int mg_model_validate(mg_model* m) {
  MG_REQUIRE(m, "mg_model_validate: null model");
  for (Prog* p : {&m->enc, &m->den, &m->dec})
    if (int rc = mg_program_validate(p->prog)) return rc;
  return 0;
}

int mg_model_vae_encode(mg_model* m, const float* rgb, float* latent, void* stream) {
  MG_REQUIRE(m && !m->host_only && rgb && latent, "mg_model_vae_encode: bad arguments (or a host-only model)");
  const hipStream_t s = (hipStream_t)stream;
  const Slot *in = m->enc.slot("rgb"), *out = m->enc.slot("latent");
  if (int rc = copy_dd(in->ptr, rgb, in->nbytes, s)) return rc;
  if (int rc = mg_program_run(m->enc.prog, stream)) return rc;
  return copy_dd(latent, out->ptr, out->nbytes, s);
}

int mg_model_denoise(mg_model* m, const float* rgb_latent, float* x, const float* step_noise, void* stream) {
  MG_REQUIRE(m && !m->host_only && rgb_latent && x, "mg_model_denoise: bad arguments (or a host-only model)");
  const hipStream_t s = (hipStream_t)stream;
  const Slot *rl = m->den.slot("rgb_latent"), *xs = m->den.slot("x");
  if (int rc = copy_dd(rl->ptr, rgb_latent, rl->nbytes, s)) return rc;
  if (int rc = copy_dd(xs->ptr, x, xs->nbytes, s)) return rc;
  const int n_noise = (int)m->hdr.cfg[8];
  MG_REQUIRE(n_noise == 0 || step_noise, "mg_model_denoise: this scheduler draws noise in %d steps: pass [%d][B][C][h][w] floats", n_noise, n_noise);
  for (int k = 0; k < n_noise; ++k) {
    char nm[24];
    snprintf(nm, sizeof(nm), "noise%d", k);
    const Slot* ns = m->den.slot(nm);
    MG_REQUIRE(ns, "mg_model_denoise: the image lacks slot %s", nm);
    if (int rc = copy_dd(ns->ptr, (const char*)step_noise + (uint64_t)k * ns->nbytes, ns->nbytes, s)) return rc;
  }
  if (int rc = mg_program_run(m->den.prog, stream)) return rc;
  return copy_dd(x, xs->ptr, xs->nbytes, s);
}

int mg_model_vae_decode(mg_model* m, const float* latent, float* pred, void* stream) {
  MG_REQUIRE(m && !m->host_only && latent && pred, "mg_model_vae_decode: bad arguments (or a host-only model)");
  const hipStream_t s = (hipStream_t)stream;
  const Slot *in = m->dec.slot("latent"), *out = m->dec.slot("pred");
  if (int rc = copy_dd(in->ptr, latent, in->nbytes, s)) return rc;
  if (int rc = mg_program_run(m->dec.prog, stream)) return rc;
  return copy_dd(pred, out->ptr, out->nbytes, s);
}

}  // extern "C"

// ---- ensemble_depth as one call (marigold/util/ensemble.py:39-196; the Python form: marigold_amd/ensemble.py::ensemble_depth) ----
namespace {
struct DevBuf {   // frees on every exit path
  void* p = nullptr;
  bool host = false;
  ~DevBuf() {
    if (p) (void)(host ? hipHostFree(p) : hipFree(p));
  }
};
mg_op make_median_op(const float* d, const float* st, float* med, float* mad, float* mm, void* scratch, int E, long long HW, int reduction,
                     int has_shift) {
  mg_op op;
  memset(&op, 0, sizeof(op));
  op.kind = MG_OP_ENS_DEPTH_MEDIAN;
  op.i[0] = E; op.i[1] = reduction; op.i[2] = has_shift;
  op.p[0] = (void*)d; op.p[1] = (void*)st; op.p[2] = med; op.p[3] = mad; op.p[4] = mm; op.p[5] = scratch;
  op.l[0] = HW;
  return op;
}
}  // namespace

extern "C" int mg_ensemble_depth(const float* preds, int E, int H, int W, int scale_invariant, int shift_invariant, int reduction,
                                 double regularizer_strength, int max_iter, double tol, int max_res, float* depth_out, float* unc_out,
                                 double* info4, void* stream) {
  MG_REQUIRE(preds && depth_out && E >= 1 && H > 0 && W > 0, "ensemble_depth: bad arguments");
  MG_REQUIRE(reduction == 0 || reduction == 1, "Unrecognized reduction method: %d.", reduction);                    // ensemble.py:86-87
  MG_REQUIRE(scale_invariant || !shift_invariant, "Pure shift-invariant ensembling is not supported.");           // :88-89
  MG_REQUIRE(scale_invariant, "Unrecognized alignment.");                                                          // :189-190
  const hipStream_t s = (hipStream_t)stream;
  const long long HW = (long long)H * W;
  const int affine = scale_invariant && shift_invariant;
  DevBuf scratch, st_dev, mm_dev;
  MG_CHECK_HIP(hipMalloc(&scratch.p, 12288));
  MG_CHECK_HIP(hipMalloc(&st_dev.p, sizeof(float) * 2 * E));
  MG_CHECK_HIP(hipMalloc(&mm_dev.p, sizeof(float) * (2 + 2 * E)));
  double fval = 0.0;
  int nit = 0, nfev = 0, status = 0;
  {
    // --- the alignment (compute_param, :154-173) on the members, down-sampled with nearest-exact beyond max_res (:158-161)
    const float* d_align = preds;
    int Ha = H, Wa = W;
    DevBuf small;
    if (max_res > 0 && (H > W ? H : W) > max_res) {
      const double f = (double)max_res / W < (double)max_res / H ? (double)max_res / W : (double)max_res / H;
      Ha = (int)(H * f);
      Wa = (int)(W * f);
      MG_CHECK_HIP(hipMalloc(&small.p, sizeof(float) * (size_t)E * Ha * Wa));
      mg_op r;
      memset(&r, 0, sizeof(r));
      r.kind = MG_OP_RESIZE;
      r.i[0] = E; r.i[1] = H; r.i[2] = W; r.i[3] = Ha; r.i[4] = Wa; r.i[5] = 2; r.i[6] = 0;
      r.p[0] = (void*)preds; r.p[1] = small.p;
      if (int rc = mg_launch(&r, stream)) return rc;
      d_align = (const float*)small.p;
    }
    const long long HWa = (long long)Ha * Wa;
    DevBuf sscratch, stats;
    MG_CHECK_HIP(hipMalloc(&sscratch.p, sizeof(double) * 128 * (size_t)E * (E + 3)));
    MG_CHECK_HIP(hipMalloc(&stats.p, sizeof(double) * (3 * (size_t)E + (size_t)E * E)));
    mg_op so;
    memset(&so, 0, sizeof(so));
    so.kind = MG_OP_ENS_DEPTH_STATS;
    so.i[0] = E;
    so.p[0] = (void*)d_align; so.p[1] = sscratch.p; so.p[2] = stats.p;
    so.l[0] = HWa;
    if (int rc = mg_launch(&so, stream)) return rc;
    std::vector<double> hs(3 * (size_t)E + (size_t)E * E);
    MG_CHECK_HIP(hipMemcpyAsync(hs.data(), stats.p, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, s));
    MG_CHECK_HIP(hipStreamSynchronize(s));
    const double *dmin = hs.data(), *dmax = dmin + E, *mean = dmax + E, *C = mean + E;
    // init_param (:163-168): fp32 arithmetic, as the reference's torch ops
    const int n = affine ? 2 * E : E;
    std::vector<double> x(n), x0;
    for (int i = 0; i < E; ++i) {
      const float lo = (float)dmin[i], hi = (float)dmax[i];
      if (affine) {
        const float rng = hi - lo;
        const float sc = 1.0f / (rng > 1e-6f ? rng : 1e-6f);
        x[i] = (double)sc;
        x[E + i] = (double)(-sc * lo);
      } else {
        x[i] = (double)(1.0f / (hi > 1e-6f ? hi : 1e-6f));
      }
    }
    x0 = x;
    // the regulariser's device pass reads its 2E parameters from, and writes its 2 + 2E results to, host-mapped memory
    DevBuf st_host, mm_host;
    st_host.host = mm_host.host = true;
    MG_CHECK_HIP(hipHostMalloc(&st_host.p, sizeof(float) * 2 * E, hipHostMallocMapped));
    MG_CHECK_HIP(hipHostMalloc(&mm_host.p, sizeof(float) * (2 + 2 * E), hipHostMallocMapped));
    void *st_d = nullptr, *mm_d = nullptr;
    MG_CHECK_HIP(hipHostGetDevicePointer(&st_d, st_host.p, 0));
    MG_CHECK_HIP(hipHostGetDevicePointer(&mm_d, mm_host.p, 0));
    const mg_op reg = make_median_op(d_align, (const float*)st_d, nullptr, nullptr, (float*)mm_d, scratch.p, E, HWa, reduction, affine);
    if (int rc = mg_ens_align_minimize(&reg, stream, E, affine, reduction, regularizer_strength, mean, C, (float*)st_host.p,
                                       (const float*)mm_host.p, x.data(), tol, max_iter, &fval, &nit, &nfev, &status))
      return rc;
    bool finite = status != 3;
    for (double v : x) finite = finite && v == v && v - v == 0.0;
    if (!finite) x = x0;   // the optimiser ended on non-finite parameters: fall back to the starting point (as the Python form)
    std::vector<float> st(2 * (size_t)E, 0.f);
    for (int i = 0; i < E; ++i) {
      st[i] = (float)x[i];
      st[E + i] = affine ? (float)x[E + i] : 0.f;
    }
    MG_CHECK_HIP(hipMemcpyAsync(st_dev.p, st.data(), sizeof(float) * st.size(), hipMemcpyHostToDevice, s));
    MG_CHECK_HIP(hipStreamSynchronize(s));   // (st is a stack-lifetime host buffer)
  }
  // --- align -> lower-middle median (+ MAD) / mean (+ std) -> min / max normalisation (:175-194), all on the device
  const mg_op fin = make_median_op(preds, (const float*)st_dev.p, depth_out, unc_out, (float*)mm_dev.p, scratch.p, E, HW, reduction, affine);
  if (int rc = mg_launch(&fin, stream)) return rc;
  mg_op no;
  memset(&no, 0, sizeof(no));
  no.kind = MG_OP_ENS_DEPTH_NORM;
  no.i[0] = affine;
  no.p[0] = depth_out; no.p[1] = unc_out; no.p[2] = mm_dev.p;
  no.l[0] = HW;
  if (int rc = mg_launch(&no, stream)) return rc;
  MG_CHECK_HIP(hipStreamSynchronize(s));   // the temporaries above are freed on return
  if (info4) { info4[0] = fval; info4[1] = nfev; info4[2] = nit; info4[3] = status; }
  return 0;
}
