// Run-time compilation of static walk programs with hipRTC.
//
// The ahead-of-time programs (static_programs.h) cover the reference's standard word sets;
// any other plan the scheduler accepts (plan.cpp, static_schedule) gets the same straight-line
// kernel here: the schedule is printed as a constexpr struct behind the device headers
// (walk_types.h, walk_scan.h, walk_device.h - embedded as text by the build, jit_sources.inc)
// and compiled for gfx950 with the flags of the ahead-of-time units (-ffp-contract=off: results
// stay bit-identical to the interpreter's).  hipRTC is loaded with dlopen: without it the plan
// simply keeps running on the interpreter.  Compiled code objects are cached on disk
// (FRUITS_HIP_JIT_CACHE, default ~/.cache/fruits_amd/jit): the file name hashes the source, the
// compile options and the hipRTC version; the file carries a header (magic, format, sizes, a
// hash of the payload) that is checked on every read; a cached object that fails to load is
// deleted and compiled again; the directory is private (0700) and only files the user owns, in a
// directory only the user can write, are trusted - a code object is executable input.
#include "jit.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <vector>

#include "launch_cache.h"

namespace fr {

namespace {

#include "jit_sources.inc"   // static const char *kJitDeviceSource

const char *kKernelExpr =
    "fr::iss_walk_static_kernel<fr::WalkCfg<2, 2, 2, 0, true, false, 4, 0, 0>, fr::JitProg>";

struct Rtc {
  void *lib = nullptr;
  int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*add_name)(void *, const char *) = nullptr;
  int (*compile)(void *, int, const char **) = nullptr;
  int (*log_size)(void *, size_t *) = nullptr;
  int (*log)(void *, char *) = nullptr;
  int (*lowered)(void *, const char *, const char **) = nullptr;
  int (*code_size)(void *, size_t *) = nullptr;
  int (*code)(void *, char *) = nullptr;
  int (*destroy)(void **) = nullptr;
  int (*version)(int *, int *) = nullptr;
  bool ok = false;
};

Rtc &rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    // FRUITS_HIP_RTC_LIB: another hipRTC (or none: tests of the interpreter fallback)
    const char *forced = getenv("FRUITS_HIP_RTC_LIB");
    if (forced && *forced) {
      r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
      }
    }
    if (!r.lib) return;
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
    r.add_name = reinterpret_cast<decltype(r.add_name)>(sym("hiprtcAddNameExpression"));
    r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
    r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
    r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
    r.lowered = reinterpret_cast<decltype(r.lowered)>(sym("hiprtcGetLoweredName"));
    r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
    r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
    r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
    r.version = reinterpret_cast<decltype(r.version)>(sym("hiprtcVersion"));
    r.ok = r.create && r.add_name && r.compile && r.log_size && r.log && r.lowered &&
           r.code_size && r.code && r.destroy;
  });
  return r;
}

unsigned long long fnv1a(const std::string &s) {
  unsigned long long h = 1469598103934665603ull;
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}

// a second, independent hash of the key (FNV-1a over the reversed string, another offset basis):
// stored in the file's header, so that a collision of the 64-bit file name - or a stale file under
// a recycled name - is refused instead of loaded as somebody else's kernel
unsigned long long fnv1a_alt(const std::string &s) {
  unsigned long long h = 0x9ae16a3b2f90404full;
  for (size_t i = s.size(); i-- > 0;) {
    h ^= (unsigned char)s[i];
    h *= 1099511628211ull;
  }
  return h;
}

std::string cache_dir() {
  if (const char *d = getenv("FRUITS_HIP_JIT_CACHE")) return *d ? std::string(d) : std::string();
  const char *home = getenv("XDG_CACHE_HOME");
  std::string base = home && *home ? std::string(home) : std::string();
  if (base.empty()) {
    const char *h = getenv("HOME");
    if (!h || !*h) return std::string();
    base = std::string(h) + "/.cache";
  }
  return base + "/fruits_amd/jit";
}

// true when `path` is a real directory (not a symbolic link) the user owns and nobody else can
// write to - checked before a cached object is written AND before one is read
bool trusted_dir(const std::string &path) {
  struct stat st;
  if (lstat(path.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// Creates the directory chain privately, then trusted_dir.
bool private_dir(const std::string &path) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0700);
  return trusted_dir(path);
}

const char *kCompileOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
constexpr int kNumCompileOptions = 4;

// what a cached object depends on besides the source text
std::string toolchain_tag() {
  std::string t;
  for (const char *o : kCompileOptions) t += std::string(o) + ";";
  Rtc &r = rtc();
  int major = 0, minor = 0;
  if (r.ok && r.version && r.version(&major, &minor) == 0)
    t += "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
  else
    t += "hiprtc ?";
  // (the HIP runtime's full version number: patch level and build included; no device needed)
  int rt = 0;
  if (hipRuntimeGetVersion(&rt) == hipSuccess) t += " hip " + std::to_string(rt);
  else (void)hipGetLastError();
  return t;
}

// Disk format: header, then the payload jit_load takes (code object, lowered kernel name, the
// name's length).
struct CacheHeader {
  char magic[8];            // "FRJITCO\1"
  uint32_t format;          // 2
  uint32_t reserved;
  uint64_t payload_bytes;
  uint64_t payload_hash;    // FNV-1a of the payload
  uint64_t key_hash;        // fnv1a_alt of what the file name hashes (source, options, toolchain)
};
constexpr uint32_t kCacheFormat = 2;
const char kCacheMagic[8] = {'F', 'R', 'J', 'I', 'T', 'C', 'O', 1};

bool parse_cached(const std::string &all, unsigned long long key_hash, std::string &payload) {
  CacheHeader h;
  if (all.size() < sizeof h) return false;
  std::memcpy(&h, all.data(), sizeof h);
  if (std::memcmp(h.magic, kCacheMagic, 8) != 0 || h.format != kCacheFormat || h.key_hash != key_hash ||
      h.payload_bytes != all.size() - sizeof h || h.payload_bytes < 8)
    return false;
  payload = all.substr(sizeof h);
  return fnv1a(payload) == h.payload_hash;
}

bool read_cached(const std::string &file, unsigned long long key_hash, std::string &payload) {
  // (the directory: the user's own and closed to others; the file: a regular one of the user's,
  // not a link)
  const size_t slash = file.rfind('/');
  if (slash == std::string::npos || !trusted_dir(file.substr(0, slash))) return false;
  struct stat st;
  if (lstat(file.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) return false;
  std::ifstream f(file, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  return parse_cached(ss.str(), key_hash, payload);
}

void write_cached(const std::string &dir, const std::string &file, unsigned long long key_hash,
                  const std::string &payload) {
  if (!private_dir(dir)) return;
  CacheHeader h{};
  std::memcpy(h.magic, kCacheMagic, 8);
  h.format = kCacheFormat;
  h.payload_bytes = payload.size();
  h.payload_hash = fnv1a(payload);
  h.key_hash = key_hash;
  const std::string tmp = file + ".tmp" + std::to_string((long long)getpid());
  std::ofstream f(tmp, std::ios::binary);
  if (!f) return;
  f.write(reinterpret_cast<const char *>(&h), sizeof h);
  f.write(payload.data(), (std::streamsize)payload.size());
  f.close();
  if (!f || rename(tmp.c_str(), file.c_str()) != 0) (void)remove(tmp.c_str());
}

std::string cache_name(const std::string &src) {
  char key[32];
  snprintf(key, sizeof key, "%016llx", fnv1a(src + "\n//" + toolchain_tag()));
  return std::string(key) + ".gfx950.co";
}

std::string cache_file(const std::string &src, std::string &dir) {
  dir = cache_dir();
  if (dir.empty()) return std::string();
  return dir + "/" + cache_name(src);
}

// The kernels SHIPPED WITH THE BUILD (fruits_amd/gen_bundle.py, run by build()): code objects
// of the fused pipelines of the reference's experiment fruits and of the BASELINE configs, in
// the cache's own file format, in `jit_bundle` next to this library (FRUITS_HIP_JIT_BUNDLE:
// another directory; empty: none).  Read-only, looked up behind the user's cache and in front
// of the compiler, so that a first launch on a fresh machine finds its own kernels.  Part of
// the installation: trusted like the library itself (the header and payload hash are checked).
std::string bundle_dir() {
  if (const char *d = getenv("FRUITS_HIP_JIT_BUNDLE")) return std::string(d);
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&bundle_dir), &info) == 0 || !info.dli_fname) return std::string();
  std::string lib(info.dli_fname);
  const size_t slash = lib.rfind('/');
  return (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/jit_bundle";
}

bool read_bundled(const std::string &name, unsigned long long key_hash, std::string &payload) {
  const std::string dir = bundle_dir();
  if (dir.empty()) return false;
  std::ifstream f(dir + "/" + name, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  return parse_cached(ss.str(), key_hash, payload);
}

}  // namespace

std::string jit_source(const StaticSchedule &sc) {
  std::ostringstream o;
  o << kJitDeviceSource << "\nnamespace fr {\nstruct JitProg {\n";
  o << "  static constexpr int groups = " << sc.groups << ", rows = " << sc.rows
    << ", frames = " << sc.frames << ";\n";
  auto list = [&](const char *name, const std::vector<int32_t> &v, size_t n) {
    o << "  static constexpr int32_t " << name << "[" << (n ? n : 1) << "] = {";
    for (size_t i = 0; i < n; ++i) o << (i ? ", " : "") << v[i];
    if (n == 0) o << 0;
    o << "};\n";
  };
  list("row_src", sc.row_src, (size_t)sc.rows);
  list("group_begin", sc.group_begin, sc.group_begin.size());
  list("group_rows", sc.group_rows, sc.group_rows.size());
  o << "  static constexpr int n = " << sc.entries.size() << ";\n";
  o << "  static constexpr int32_t w[" << sc.entries.size() * 16 << "] = {\n";
  for (const NodeRec &r : sc.entries) {
    o << "     ";
    for (int i = 0; i < 16; ++i) o << " " << r.w[i] << ",";
    o << "\n";
  }
  o << "  };\n};\n}  // namespace fr\n";
  return o.str();
}

// Compiles `src` for gfx950 and returns the payload jit_load takes (code object, lowered name of
// `kernel_expr`, the name's length); served from the disk cache when it holds a valid object for
// this source, toolchain and option set.
// `cache_only`: take the code object from the disk cache or fail (err = kNotCached) - nothing is
// compiled
static const char *const kNotCached = "not in the disk cache";
// `into`: write the code object THERE (a bundle being built, fruits_amd/gen_bundle.py) instead
// of into the user's cache; what that directory already holds is not compiled again
static bool compile_source(const std::string &src, const char *kernel_expr, const char *const *opts,
                           int n_opts, std::string &code, std::string &err, bool *from_cache,
                           bool cache_only = false, const char *into = nullptr, bool no_bundle = false) {
  if (from_cache) *from_cache = false;
  std::string keyed = src + "\n//" + kernel_expr;
  for (int i = 0; i < n_opts; ++i) keyed += std::string(" ") + opts[i];
  const unsigned long long key_hash = fnv1a_alt(keyed + "\n//" + toolchain_tag());
  std::string dir;
  std::string file = cache_file(keyed, dir);
  if (into != nullptr) {
    dir = into;
    file = dir + "/" + cache_name(keyed);
  }
  if (!file.empty()) {
    if (read_cached(file, key_hash, code)) {
      if (from_cache) *from_cache = true;
      // (a bundle being rebuilt: what is still wanted is marked by its time stamp, the rest is
      // swept afterwards - fruits_amd/gen_bundle.py)
      if (into != nullptr) (void)utimensat(AT_FDCWD, file.c_str(), nullptr, 0);
      return true;
    }
    code.clear();
  }
  if (into == nullptr && !no_bundle && read_bundled(cache_name(keyed), key_hash, code)) {
    if (from_cache) *from_cache = true;
    return true;
  }
  code.clear();
  if (cache_only) {
    err = kNotCached;
    return false;
  }
  Rtc &r = rtc();
  if (!r.ok) {
    err = "hipRTC (libhiprtc.so) is not available";
    return false;
  }
  void *prog = nullptr;
  if (r.create(&prog, src.c_str(), "fruits_static_program.hip", 0, nullptr, nullptr) != 0) {
    err = "hiprtcCreateProgram failed";
    return false;
  }
  bool ok = false;
  if (r.add_name(prog, kernel_expr) != 0) {
    err = "hiprtcAddNameExpression failed";
  } else if (r.compile(prog, n_opts, const_cast<const char **>(opts)) != 0) {
    size_t n = 0;
    r.log_size(prog, &n);
    std::string log(n, '\0');
    if (n) r.log(prog, log.data());
    err = "hipRTC compilation failed:\n" + log.substr(0, 4000);
  } else {
    const char *low = nullptr;
    size_t n = 0;
    if (r.lowered(prog, kernel_expr, &low) != 0 || !low || r.code_size(prog, &n) != 0 || n == 0) {
      err = "hipRTC returned no code";
    } else {
      // the code object, then the lowered kernel name (the loader needs it)
      std::string co(n, '\0');
      r.code(prog, co.data());
      const std::string name(low);
      code = co;
      code.append(name);
      const unsigned int len = (unsigned int)name.size();
      code.append(reinterpret_cast<const char *>(&len), 4);
      ok = true;
    }
  }
  r.destroy(&prog);
  if (ok && !file.empty()) write_cached(dir, file, key_hash, code);
  return ok;
}

static void drop_cached(const std::string &src, const char *kernel_expr, const char *const *opts, int n_opts) {
  std::string keyed = src + "\n//" + kernel_expr;
  for (int i = 0; i < n_opts; ++i) keyed += std::string(" ") + opts[i];
  std::string dir;
  const std::string file = cache_file(keyed, dir);
  if (!file.empty()) (void)remove(file.c_str());
}

// Loads a payload: module + function of the kernel it names.
static bool load_payload(const std::string &code, hipModule_t &mod, hipFunction_t &fn, std::string &err) {
  if (code.size() < 8) {
    err = "empty code object";
    return false;
  }
  unsigned int len = 0;
  std::memcpy(&len, code.data() + code.size() - 4, 4);
  if ((size_t)len + 4 > code.size()) {
    err = "corrupt code object";
    return false;
  }
  const std::string name = code.substr(code.size() - 4 - len, len);
  const std::string image = code.substr(0, code.size() - 4 - len);
  mod = nullptr;
  fn = nullptr;
  if (hipModuleLoadData(&mod, image.data()) != hipSuccess ||
      hipModuleGetFunction(&fn, mod, name.c_str()) != hipSuccess) {
    (void)hipGetLastError();
    if (mod) (void)hipModuleUnload(mod);
    mod = nullptr;
    err = "hipModuleLoadData failed for the compiled program";
    return false;
  }
  return true;
}

void jit_cache_drop(const StaticSchedule &sc) {
  if (!sc.ok) return;
  drop_cached(jit_source(sc), kKernelExpr, kCompileOptions, kNumCompileOptions);
}

bool jit_compile(const StaticSchedule &sc, std::string &code, std::string &err, bool *from_cache) {
  if (from_cache) *from_cache = false;
  if (!sc.ok) {
    err = "the plan has no static schedule";
    return false;
  }
  return compile_source(jit_source(sc), kKernelExpr, kCompileOptions, kNumCompileOptions, code, err,
                        from_cache);
}

// ---------------------------------------------------------------- fused walk, ops as constants
// The fused walk of ONE pipeline (walk_fused.h): the kernel instantiation its plan and series
// length select, over a JitOps struct that holds the sieves' kind / differencing order / shape /
// cuts as immediates.  Same flags as the ahead-of-time fused units (fruits_amd/build.py).
static const char *kFusedOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                      "-mllvm", "-structurizecfg-skip-uniform-regions"};
constexpr int kNumFusedOptions = 6;

static std::string fused_kernel_expr(const FusedKey &k, bool with_plan) {
  std::ostringstream o;
  o << "fr::iss_fused_kernel<fr::WalkCfg<" << k.E << ", 1, " << k.LV << ", " << k.MULTI << ", true, "
    << (k.W ? "true" : "false") << ", 4, 1, " << k.SEMI << ", false, " << (k.TI ? "true" : "false")
    << ", " << (k.HO ? "true" : "false") << ">, " << (k.TOTAL ? "true" : "false") << ", fr::JitOps"
    << (with_plan ? ", fr::JitPlan>" : ">");
  return o.str();
}

std::string jit_fused_source(const FusedOps &ops, const FusedPlan *plan) {
  std::ostringstream o;
  o << kJitDeviceSource << "\nnamespace fr {\nstruct JitOps {\n  static constexpr bool is_static = true;\n";
  const size_t n = ops.w0.size();
  o << "  static constexpr int n = " << n << ", n_padded = " << ops.n_padded << ", cps = " << ops.cps
    << ";\n  static constexpr bool window_fits = " << (ops.window_fits ? "true" : "false")
    << ", full_chunks = " << (ops.full_chunks ? "true" : "false") << ";\n";
  auto list = [&](const char *name, const std::vector<int32_t> &v) {
    o << "  static constexpr int32_t " << name << "[" << (n ? n : 1) << "] = {";
    for (size_t i = 0; i < n; ++i) o << (i ? ", " : "") << v[i];
    if (n == 0) o << 0;
    o << "};\n";
  };
  list("w0", ops.w0);
  list("lo", ops.lo);
  list("hi", ops.hi);
  o << "};\n";
  if (plan != nullptr && plan->piece) {
    o << "struct JitPlan {\n  static constexpr bool is_static = false, is_shaped = false, is_piece = true;\n"
      << "  static constexpr int32_t w[" << plan->w.size() << "] = {";
    for (size_t i = 0; i < plan->w.size(); ++i) o << (i ? (i % 16 ? ", " : ",\n    ") : "\n    ") << plan->w[i];
    o << "};\n};\n";
  } else if (plan != nullptr && !plan->shapes.empty()) {
    o << "struct JitPlan {\n  static constexpr bool is_static = false, is_shaped = true, is_piece = false;\n"
      << "  static constexpr int n = " << plan->shapes.size() << ";\n  static constexpr int32_t desc["
      << plan->shapes.size() << "] = {";
    for (size_t i = 0; i < plan->shapes.size(); ++i) o << (i ? ", " : "") << plan->shapes[i];
    o << "};\n};\n";
  } else if (plan != nullptr) {
    o << "struct JitPlan {\n  static constexpr bool is_static = true, is_shaped = false, is_piece = false;\n  static constexpr int groups = "
      << plan->groups() << ";\n  static constexpr int32_t group_begin[" << plan->groups() << "] = {";
    for (int g = 0; g < plan->groups(); ++g) o << (g ? ", " : "") << plan->group_begin[g];
    o << "};\n  static constexpr int32_t w[" << plan->w.size() << "] = {";
    for (size_t i = 0; i < plan->w.size(); ++i) o << (i ? (i % 16 ? ", " : ",\n    ") : "\n    ") << plan->w[i];
    o << "};\n};\n";
  }
  o << "}  // namespace fr\n";
  return o.str();
}

bool jit_not_cached(const std::string &err) { return err == kNotCached; }

bool jit_fused_into(const FusedOps &ops, const FusedKey &key, const FusedPlan *plan, const char *dir,
                    std::string &err) {
  if (ops.w0.empty() || ops.w0.size() > 64) {
    err = "no ops (or more than 64) per output row";
    return false;
  }
  const std::string src = jit_fused_source(ops, plan), expr = fused_kernel_expr(key, plan != nullptr);
  std::string code;
  return compile_source(src, expr.c_str(), kFusedOptions, kNumFusedOptions, code, err, nullptr, false, dir);
}

bool jit_fused(const FusedOps &ops, const FusedKey &key, JitProgram &out, std::string &err,
               const FusedPlan *plan, bool cache_only) {
  if (ops.w0.empty() || ops.w0.size() > 64) {
    err = "no ops (or more than 64) per output row";
    return false;
  }
  const std::string src = jit_fused_source(ops, plan), expr = fused_kernel_expr(key, plan != nullptr);
  for (int attempt = 0; attempt < 2; ++attempt) {
    std::string code;
    bool from_cache = false;
    // (second attempt: the first one's object - from the user's cache, dropped below, or shipped
    // with the build - was refused by the loader)
    if (!compile_source(src, expr.c_str(), kFusedOptions, kNumFusedOptions, code, err, &from_cache,
                        cache_only, nullptr, attempt > 0))
      return false;
    hipModule_t mod;
    hipFunction_t fn;
    if (load_payload(code, mod, fn, err)) {
      out = JitProgram{};
      out.module = mod;
      out.fn = fn;
      out.device = current_device();
      out.groups = plan != nullptr ? plan->groups() : 0;
      return true;
    }
    if (!from_cache) return false;
    // a cached object the loader refuses (another ROCm, a damaged file): compile afresh, once
    drop_cached(src, expr.c_str(), kFusedOptions, kNumFusedOptions);
    if (cache_only) {
      err = kNotCached;
      return false;
    }
  }
  return false;
}

hipError_t jit_launch_fused(const JitProgram &p, const IssArgs &a, size_t lds_bytes, hipStream_t st) {
  const int64_t units = a.N * a.G;
  if (units <= 0) return hipSuccess;
  if (units > 0x7fffffffLL || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  IssArgs args = a;
  void *params[] = {&args};
  return hipModuleLaunchKernel(p.fn, (unsigned)units, 1, 1, kWalkThreads, 1, 1, (unsigned)lds_bytes, st,
                               params, nullptr);
}

bool jit_load(const std::string &code, const StaticSchedule &sc, JitProgram &out, std::string &err) {
  hipModule_t mod;
  hipFunction_t fn;
  if (!load_payload(code, mod, fn, err)) return false;
  out.module = mod;
  out.fn = fn;
  out.groups = sc.groups;
  out.lds_bytes = ((size_t)sc.rows * 1024 + 4 * 4) * sizeof(double);
  out.device = current_device();
  int nb = 0;
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kWalkThreads, out.lds_bytes) !=
          hipSuccess || nb < 1) {
    (void)hipGetLastError();
    nb = 1;
  }
  out.per_cu = nb;
  return true;
}

void jit_unload(JitProgram &p) {
  if (p.module) (void)hipModuleUnload(p.module);
  p = JitProgram{};
}

hipError_t jit_launch(const JitProgram &p, const IssArgs &a, hipStream_t st) {
  const int64_t units = a.N * p.groups;
  if (units <= 0) return hipSuccess;
  if (units > 0x7fffffffLL || a.G != p.groups) return hipErrorInvalidValue;
  // (the resident count of the grid as launched: with the launch's own LDS pad)
  int per_cu = p.per_cu;
  if (a.lds_pad != 0 && a.persistent) {
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, p.fn, kWalkThreads,
                                                           p.lds_bytes + (size_t)a.lds_pad) == hipSuccess && nb >= 1)
      per_cu = nb;
    else
      (void)hipGetLastError();
  }
  int64_t resident = (int64_t)per_cu * device_cu_count();
  resident -= resident % 8;
  if (resident < 8) resident = 8;
  const int64_t blocks = (units < resident || !a.persistent) ? units : resident;
  IssArgs args = a;
  void *params[] = {&args};
  return hipModuleLaunchKernel(p.fn, (unsigned)blocks, 1, 1, kWalkThreads, 1, 1,
                               (unsigned)(p.lds_bytes + (size_t)a.lds_pad), st, params, nullptr);
}

}  // namespace fr
