// gvl_host.hip -- host-only entry points of libgvl.so that launch no kernels of their own: the packed-weight file reader
// (gvl_load_packed: safetensors container -> gvl_load_weight), the KV-pool query, and the RCCL exchange of visual tokens
// (gvl_comm_* / gvl_allgather_visual; librccl is dlopen'ed so that libgvl.so itself links only the HIP runtime).
// Reference counterparts: torch.load + load_state_dict (inference.py:156-162, models/llava_next_video.py:117-151); the exchange has
// none (the reference's inference is single-GPU, inference.py:17; SURVEY.md §8 e).
#include "gvl_ctx.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace { inline int fail(gvl_ctx* c, int code, const std::string& msg) { return gvl_fail(c, code, msg); } }

extern "C" {

int gvl_kv_info(const gvl_ctx* ctx, int* total_pages, int* free_pages, int64_t* pool_bytes, int* max_live_seqs) {
  if (!ctx) return GVL_ERR_ARG;
  if (total_pages) *total_pages = ctx->kv_total_pages;
  if (free_pages) *free_pages = (int)ctx->free_pages.size();
  if (pool_bytes) *pool_bytes = (int64_t)(ctx->layer_stride * ctx->cfg.layers * 2 * 2);
  if (max_live_seqs) *max_live_seqs = gvl_ctx::kMaxSeqs;
  return 0;
}

int gvl_decode_group_info(const gvl_ctx* ctx, int* max_group, int* any_size) {
  if (!ctx) return GVL_ERR_ARG;
  if (max_group) *max_group = ctx->decode_mfma ? GVL_MAX_DECODE_BATCH : GVL_MAX_VALU_BATCH;
  if (any_size) *any_size = ctx->decode_mfma ? 1 : 0;
  return 0;
}

// ---- packed weight file (safetensors container): u64 header length, JSON header, raw little-endian tensor bytes ----------------
namespace {
struct StEntry { std::string name, dtype; std::vector<int64_t> shape; uint64_t b = 0, e = 0; };
// Minimal reader for the restricted JSON a safetensors header is: {"name": {"dtype": "...", "shape": [..], "data_offsets": [b, e]}, ...,
// "__metadata__": {"k": "v", ...}}.  Returns false on anything else.
struct StParser {
  const char* p; const char* end; std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
  bool str(std::string& out) {
    ws(); if (p >= end || *p != '"') return false; ++p; out.clear();
    while (p < end && *p != '"') {
      if (*p == '\\') { if (p + 1 >= end) return false; const char c = p[1]; p += 2;
        if (c == 'u') { if (p + 4 > end) return false; out += '?'; p += 4; } else out += (c == 'n' ? '\n' : c == 't' ? '\t' : c); }
      else out += *p++;
    }
    if (p >= end) return false; ++p; return true;
  }
  bool num(uint64_t& v) { ws(); if (p >= end || *p < '0' || *p > '9') return false; v = 0; while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (uint64_t)(*p++ - '0'); return true; }
  bool numlist(std::vector<int64_t>& v) {
    v.clear(); if (!lit('[')) return false; if (lit(']')) return true;
    for (;;) { uint64_t x; if (!num(x)) return false; v.push_back((int64_t)x); if (lit(']')) return true; if (!lit(',')) return false; }
  }
  bool parse(std::vector<StEntry>& out, std::unordered_map<std::string, std::string>& meta) {
    if (!lit('{')) return false; if (lit('}')) return true;
    for (;;) {
      std::string key; if (!str(key) || !lit(':') || !lit('{')) return false;
      if (key == "__metadata__") {
        if (!lit('}')) for (;;) { std::string k, v; if (!str(k) || !lit(':') || !str(v)) return false; meta[k] = v; if (lit('}')) break; if (!lit(',')) return false; }
      } else {
        StEntry en; en.name = key; bool have_off = false;
        for (;;) {
          std::string k; if (!str(k) || !lit(':')) return false;
          if (k == "dtype") { if (!str(en.dtype)) return false; }
          else if (k == "shape") { if (!numlist(en.shape)) return false; }
          else if (k == "data_offsets") { std::vector<int64_t> o; if (!numlist(o) || o.size() != 2) return false; en.b = (uint64_t)o[0]; en.e = (uint64_t)o[1]; have_off = true; }
          else return false;
          if (lit('}')) break; if (!lit(',')) return false;
        }
        if (!have_off) return false;
        out.push_back(en);
      }
      if (lit('}')) return true; if (!lit(',')) return false;
    }
  }
};
}  // namespace

int gvl_load_packed(gvl_ctx* ctx, const char* path, int* n_loaded) {
  if (!ctx || !path) return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: bad argument");
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(ctx, GVL_ERR_ARG, std::string("gvl_load_packed: cannot open ") + path);
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size < 8) { close(fd); return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: file too short"); }
  const size_t fsize = (size_t)sb.st_size;
  void* map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: mmap failed");
  struct Unmap { void* m; size_t n; ~Unmap() { munmap(m, n); } } unmap{map, fsize};
  const unsigned char* base = (const unsigned char*)map;
  uint64_t hlen = 0; for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | base[i];
  if (hlen > fsize - 8) return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: header length exceeds the file");
  std::vector<StEntry> ents; std::unordered_map<std::string, std::string> meta;
  StParser ps{(const char*)base + 8, (const char*)base + 8 + hlen, {}};
  if (!ps.parse(ents, meta)) return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: malformed safetensors header");
  if (meta["format"] != "gvl-packed-1") return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: not a gvl packed weight file (metadata format != gvl-packed-1)");
  const unsigned char* data = base + 8 + hlen; const size_t dsize = fsize - 8 - hlen;
  int n = 0;
  for (const StEntry& en : ents) {
    int dt; size_t esz;
    if (en.dtype == "BF16") { dt = GVL_BF16; esz = 2; } else if (en.dtype == "F32") { dt = GVL_F32; esz = 4; }
    else return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: tensor " + en.name + " has dtype " + en.dtype + " (want BF16 / F32)");
    int64_t numel = 1; for (int64_t d : en.shape) numel *= d;
    if (en.e < en.b || en.e > dsize || (uint64_t)numel * esz != en.e - en.b || en.shape.size() > 8) return fail(ctx, GVL_ERR_ARG, "gvl_load_packed: bad offsets / shape for " + en.name);
    int64_t one = 1;
    const int rc = gvl_load_weight(ctx, en.name.c_str(), data + en.b, dt, en.shape.empty() ? &one : en.shape.data(), en.shape.empty() ? 1 : (int)en.shape.size(), 0);
    if (rc) return rc;
    ++n;
  }
  if (n_loaded) *n_loaded = n;
  return 0;
}

// ---- RCCL (dlopen'ed: libgvl.so itself links only the HIP runtime) ------------------------------------------------------------
namespace {
struct Rccl {
  struct UID { char b[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
  void* h = nullptr; bool tried = false; std::string err;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, UID, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;     // ncclBroadcast(send, recv, count, type, root, comm, stream)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} g_rccl;
bool rccl_load() {
  if (g_rccl.tried) return g_rccl.h != nullptr;
  g_rccl.tried = true;
  // GVL_RCCL_LIB names the library explicitly (a deployment knob like NCCL's own; also how the host test reaches the failure path)
  const char* over = getenv("GVL_RCCL_LIB");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  std::string last = "?";
  if (over && *over) {
    g_rccl.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) { const char* e = dlerror(); if (e) last = e; }   // dlerror() clears the pending message: read it ONCE
  } else {
    for (const char* n : names) {
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.h) break;
      const char* e = dlerror(); if (e) last = e;
    }
  }
  if (!g_rccl.h) { g_rccl.err = std::string("dlopen(librccl) failed: ") + last; return false; }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.h, "ncclAllGather");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.h, "ncclCommCount");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))dlsym(g_rccl.h, "ncclBroadcast");        // optional: gvl_allgatherv_visual only
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.h, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.h, "ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) { g_rccl.err = "librccl lacks a required symbol"; dlclose(g_rccl.h); g_rccl.h = nullptr; return false; }
  return true;
}
std::string rccl_msg(const char* what, int rc) { return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error") + " (" + std::to_string(rc) + ")"; }
constexpr int kNcclBfloat16 = 9;    // ncclDataType_t, rccl.h
}  // namespace

int gvl_comm_unique_id(char id_out[128]) {
  if (!id_out) return fail(nullptr, GVL_ERR_ARG, "gvl_comm_unique_id: null");
  if (!rccl_load()) return fail(nullptr, GVL_ERR_STATE, g_rccl.err);
  const int rc = g_rccl.GetUniqueId(id_out);
  if (rc) return fail(nullptr, GVL_ERR_HIP, rccl_msg("ncclGetUniqueId", rc));
  return 0;
}
int gvl_comm_init(gvl_ctx* ctx, const char id[128], int rank, int world) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail(ctx, GVL_ERR_ARG, "gvl_comm_init: bad arguments");
  if (ctx->comm) return fail(ctx, GVL_ERR_STATE, "gvl_comm_init: communicator already initialised");
  if (!rccl_load()) return fail(ctx, GVL_ERR_STATE, g_rccl.err);
  Rccl::UID uid; memcpy(uid.b, id, 128);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, world, uid, rank);
  if (rc) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclCommInitRank", rc));
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
  return 0;
}
int gvl_comm_destroy(gvl_ctx* ctx) {
  if (!ctx) return GVL_ERR_ARG;
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
  ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0;
  return 0;
}
int gvl_comm_count(gvl_ctx* ctx, int* n_ranks) {
  if (!ctx || !n_ranks) return fail(ctx, GVL_ERR_ARG, "gvl_comm_count: null");
  if (!ctx->comm) { *n_ranks = 1; return 0; }      // no communicator: a single-rank job
  if (!rccl_load() || !g_rccl.CommCount) return fail(ctx, GVL_ERR_STATE, g_rccl.CommCount ? g_rccl.err : "librccl lacks ncclCommCount");
  const int rc = g_rccl.CommCount(ctx->comm, n_ranks);
  if (rc) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclCommCount", rc));
  return 0;
}
int gvl_allgather_visual(gvl_ctx* ctx, void* comm, const uint16_t* local, int rows_per_rank, int hidden, uint16_t* all, void* stream) {
  if (!ctx || !local || !all || rows_per_rank <= 0 || hidden <= 0) return fail(ctx, GVL_ERR_ARG, "gvl_allgather_visual: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  void* cm = comm ? comm : ctx->comm;
  const size_t count = (size_t)rows_per_rank * hidden;
  if (!cm) {   // no communicator: a single-rank job
    if (local != all) HIPCHK(ctx, hipMemcpyAsync(all, local, count * 2, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (!rccl_load()) return fail(ctx, GVL_ERR_STATE, g_rccl.err);
  const int rc = g_rccl.AllGather(local, all, count, kNcclBfloat16, cm, st);
  if (rc) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclAllGather", rc));
  return 0;
}

// Uneven blocks (12 segments over 8 ranks: 2,2,2,2,1,1,1,1) straight into the segment-ordered prefix: rank r's rows land at row offset sum(rows[0..r)) of
// `all` -- no padding to the largest block, no re-assembly copy afterwards.  One ncclGroup of `world` broadcasts (root r sends its block, everybody
// receives it in place): the standard all-gather-v; on xGMI every root pushes to all peers at once.
int gvl_allgatherv_visual(gvl_ctx* ctx, void* comm, const uint16_t* local, const int* rows_per_rank, int hidden, uint16_t* all, void* stream) {
  if (!ctx || !all || !rows_per_rank || hidden <= 0) return fail(ctx, GVL_ERR_ARG, "gvl_allgatherv_visual: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  void* cm = comm ? comm : ctx->comm;
  const int world = cm ? ctx->comm_world : 1, rank = cm ? ctx->comm_rank : 0;
  if (comm && comm != ctx->comm) return fail(ctx, GVL_ERR_ARG, "gvl_allgatherv_visual: a foreign communicator's rank / size are unknown here -- pass NULL (the ctx's own)");
  size_t off = 0, my_off = 0;
  for (int r = 0; r < world; ++r) { if (rows_per_rank[r] < 0) return fail(ctx, GVL_ERR_ARG, "gvl_allgatherv_visual: negative block"); if (r == rank) my_off = off; off += (size_t)rows_per_rank[r] * hidden; }
  if (rows_per_rank[rank] > 0 && !local) return fail(ctx, GVL_ERR_ARG, "gvl_allgatherv_visual: null local block");
  if (!cm) {
    if (rows_per_rank[0] > 0 && local != all) HIPCHK(ctx, hipMemcpyAsync(all, local, (size_t)rows_per_rank[0] * hidden * 2, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (!rccl_load()) return fail(ctx, GVL_ERR_STATE, g_rccl.err);
  if (!g_rccl.Broadcast || !g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(ctx, GVL_ERR_STATE, "librccl lacks ncclBroadcast / ncclGroupStart / ncclGroupEnd");
  int rc = g_rccl.GroupStart();
  if (rc) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclGroupStart", rc));
  size_t o = 0;
  for (int r = 0; r < world && !rc; ++r) {
    const size_t cnt = (size_t)rows_per_rank[r] * hidden;
    if (cnt) rc = g_rccl.Broadcast(r == rank ? (const void*)local : (const void*)(all + o), all + o, cnt, kNcclBfloat16, r, cm, st);
    o += cnt;
  }
  const int rc2 = g_rccl.GroupEnd();
  (void)my_off;
  if (rc) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclBroadcast", rc));
  if (rc2) return fail(ctx, GVL_ERR_HIP, rccl_msg("ncclGroupEnd", rc2));
  return 0;
}

}  // extern "C"
