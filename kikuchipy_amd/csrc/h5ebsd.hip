// h5ebsd.hip - direct reader of kikuchipy's h5ebsd files (host code only)
//   io/plugins/kikuchipy_h5ebsd/_api.py:64-160   scan2dict: header fields, patterns, static background, PCs
//   io/plugins/_h5ebsd.py:303-390                get_data: (ny, nx, sy, sx) reshape, zero padding of short files
// so that the engine can take `Scan N/EBSD/Data/patterns` and `Header/static_background`
// from a file without HyperSpy / h5py: the HDF5 C library is dlopen()ed at run time (like
// RCCL), and kpdi_set_experimental_h5ebsd reads the patterns into a pinned host buffer from
// which they go to the device with one DMA transfer.
#include "../../include/kpdi.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace kpdi {
int fail_msg(int code, const char *fmt, ...);  // api.hip: records the thread's last error
}

namespace {

typedef int64_t hid_t;  // HDF5 >= 1.10
typedef int herr_t;
typedef unsigned long long hsize_t;
typedef int htri_t;

struct Hdf5 {
  void *lib = nullptr;
  std::string path;
  herr_t (*open)() = nullptr;
  hid_t (*Fopen)(const char *, unsigned, hid_t) = nullptr;
  herr_t (*Fclose)(hid_t) = nullptr;
  htri_t (*Lexists)(hid_t, const char *, hid_t) = nullptr;
  long (*Lget_name_by_idx)(hid_t, const char *, int, int, hsize_t, char *, size_t, hid_t) = nullptr;
  hid_t (*Dopen2)(hid_t, const char *, hid_t) = nullptr;
  herr_t (*Dclose)(hid_t) = nullptr;
  hid_t (*Dget_space)(hid_t) = nullptr;
  hid_t (*Dget_type)(hid_t) = nullptr;
  herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *) = nullptr;
  herr_t (*Sclose)(hid_t) = nullptr;
  int (*Sget_simple_extent_ndims)(hid_t) = nullptr;
  int (*Sget_simple_extent_dims)(hid_t, hsize_t *, hsize_t *) = nullptr;
  herr_t (*Tclose)(hid_t) = nullptr;
  int (*Tget_class)(hid_t) = nullptr;
  size_t (*Tget_size)(hid_t) = nullptr;
  int (*Tget_sign)(hid_t) = nullptr;
  hid_t (*Tget_native_type)(hid_t, int) = nullptr;
  herr_t (*Eset_auto2)(hid_t, void *, void *) = nullptr;

  bool load() {
    if (lib) return true;
    std::vector<std::string> names;
    if (const char *env = getenv("KPDI_HDF5_LIB")) names.push_back(env);
    for (const char *n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310",
                          "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"})
      names.push_back(n);
    for (const auto &n : names) {
      lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (lib) {
        path = n;
        break;
      }
    }
    if (!lib) return false;
#define KPDI_H5(field, name)                     \
  field = (decltype(field))dlsym(lib, name);     \
  if (!field) {                                  \
    dlclose(lib);                                \
    lib = nullptr;                               \
    return false;                                \
  }
    KPDI_H5(open, "H5open")
    KPDI_H5(Fopen, "H5Fopen")
    KPDI_H5(Fclose, "H5Fclose")
    KPDI_H5(Lexists, "H5Lexists")
    KPDI_H5(Lget_name_by_idx, "H5Lget_name_by_idx")
    KPDI_H5(Dopen2, "H5Dopen2")
    KPDI_H5(Dclose, "H5Dclose")
    KPDI_H5(Dget_space, "H5Dget_space")
    KPDI_H5(Dget_type, "H5Dget_type")
    KPDI_H5(Dread, "H5Dread")
    KPDI_H5(Sclose, "H5Sclose")
    KPDI_H5(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims")
    KPDI_H5(Sget_simple_extent_dims, "H5Sget_simple_extent_dims")
    KPDI_H5(Tclose, "H5Tclose")
    KPDI_H5(Tget_class, "H5Tget_class")
    KPDI_H5(Tget_size, "H5Tget_size")
    KPDI_H5(Tget_sign, "H5Tget_sign")
    KPDI_H5(Tget_native_type, "H5Tget_native_type")
    KPDI_H5(Eset_auto2, "H5Eset_auto2")
#undef KPDI_H5
    open();
    Eset_auto2(0, nullptr, nullptr);  // errors are reported through kpdi_last_error, not stderr
    return true;
  }
};
Hdf5 g_h5;

constexpr int H5T_INTEGER = 0, H5T_FLOAT = 1;
constexpr int H5T_SGN_NONE = 0;
constexpr int H5T_DIR_ASCEND = 1;

struct Dataset {
  hid_t id = -1, space = -1, type = -1, native = -1;
  std::vector<hsize_t> dims;
  int cls = -1, sign = 0;
  size_t esize = 0;
  size_t count() const {
    size_t n = 1;
    for (hsize_t d : dims) n *= (size_t)d;
    return n;
  }
  void close() {
    if (native >= 0) g_h5.Tclose(native);
    if (type >= 0) g_h5.Tclose(type);
    if (space >= 0) g_h5.Sclose(space);
    if (id >= 0) g_h5.Dclose(id);
    id = space = type = native = -1;
  }
};

bool open_dataset(hid_t file, const std::string &name, Dataset *d) {
  // H5Lexists needs every intermediate group to exist
  size_t pos = 0;
  while ((pos = name.find('/', pos + 1)) != std::string::npos)
    if (g_h5.Lexists(file, name.substr(0, pos).c_str(), 0) <= 0) return false;
  if (g_h5.Lexists(file, name.c_str(), 0) <= 0) return false;
  d->id = g_h5.Dopen2(file, name.c_str(), 0);
  if (d->id < 0) return false;
  d->space = g_h5.Dget_space(d->id);
  d->type = g_h5.Dget_type(d->id);
  d->native = g_h5.Tget_native_type(d->type, H5T_DIR_ASCEND);
  const int nd = g_h5.Sget_simple_extent_ndims(d->space);
  d->dims.resize(nd > 0 ? nd : 0);
  if (nd > 0) g_h5.Sget_simple_extent_dims(d->space, d->dims.data(), nullptr);
  d->cls = g_h5.Tget_class(d->type);
  d->esize = g_h5.Tget_size(d->native);
  d->sign = d->cls == H5T_INTEGER ? g_h5.Tget_sign(d->type) : 0;
  return true;
}

int kpdi_dtype_of(const Dataset &d) {
  if (d.cls == H5T_INTEGER) {
    const bool u = d.sign == H5T_SGN_NONE;
    switch (d.esize) {
      case 1: return u ? KPDI_U8 : KPDI_I8;
      case 2: return u ? KPDI_U16 : KPDI_I16;
      case 4: return u ? KPDI_U32 : KPDI_I32;
    }
  } else if (d.cls == H5T_FLOAT) {
    if (d.esize == 4) return KPDI_F32;
    if (d.esize == 8) return KPDI_F64;
  }
  return -1;
}

// a header value stored as a one-element dataset (io/plugins/_h5ebsd.py `_hdf5group2dict`)
bool read_scalar(hid_t file, const std::string &name, double *out) {
  Dataset d;
  if (!open_dataset(file, name, &d)) return false;
  bool ok = false;
  if (d.count() == 1 && (d.cls == H5T_INTEGER || d.cls == H5T_FLOAT) && d.esize <= 8) {
    unsigned char buf[8] = {0};
    if (g_h5.Dread(d.id, d.native, 0, 0, 0, buf) >= 0) {
      ok = true;
      if (d.cls == H5T_FLOAT)
        *out = d.esize == 4 ? (double)*(float *)buf : *(double *)buf;
      else if (d.sign == H5T_SGN_NONE)
        *out = d.esize == 1 ? (double)*(uint8_t *)buf : d.esize == 2 ? (double)*(uint16_t *)buf
               : d.esize == 4 ? (double)*(uint32_t *)buf : (double)*(uint64_t *)buf;
      else
        *out = d.esize == 1 ? (double)*(int8_t *)buf : d.esize == 2 ? (double)*(int16_t *)buf
               : d.esize == 4 ? (double)*(int32_t *)buf : (double)*(int64_t *)buf;
    }
  }
  d.close();
  return ok;
}

struct File {
  hid_t id = -1;
  std::string scan;
  ~File() {
    if (id >= 0) g_h5.Fclose(id);
  }
};

// open `path` and settle on the scan group: the one named, or the first "Scan ..." group
int open_scan(const char *path, const char *scan, File *f) {
  if (!path) return kpdi::fail_msg(KPDI_EINVAL, "file path is NULL");
  if (!g_h5.load())
    return kpdi::fail_msg(KPDI_EINVAL, "the HDF5 C library could not be loaded (tried libhdf5.so, /opt/conda/lib/libhdf5.so; "
                                       "set KPDI_HDF5_LIB): %s", dlerror() ? dlerror() : "symbol missing");
  f->id = g_h5.Fopen(path, 0 /* H5F_ACC_RDONLY */, 0);
  if (f->id < 0) return kpdi::fail_msg(KPDI_EINVAL, "cannot open '%s' as an HDF5 file", path);
  if (scan && *scan) {
    f->scan = scan;
  } else {
    char name[256];
    for (hsize_t i = 0;; ++i) {
      const long n = g_h5.Lget_name_by_idx(f->id, ".", 0 /* H5_INDEX_NAME */, 0 /* H5_ITER_INC */, i, name, sizeof name, 0);
      if (n < 0) break;
      if (strncmp(name, "Scan", 4) == 0) {
        f->scan = name;
        break;
      }
    }
    if (f->scan.empty()) return kpdi::fail_msg(KPDI_EINVAL, "'%s' holds no 'Scan ...' group", path);
  }
  if (g_h5.Lexists(f->id, f->scan.c_str(), 0) <= 0)
    return kpdi::fail_msg(KPDI_EINVAL, "Scan '%s' is not among the scans of '%s'", f->scan.c_str(), path);
  return KPDI_OK;
}

int fill_info(File &f, kpdi_h5ebsd_info *info) {
  memset(info, 0, sizeof *info);
  const std::string h = f.scan + "/EBSD/Header/";
  double v = 0;
  auto need = [&](const char *key, int32_t *dst) {
    if (!read_scalar(f.id, h + key, &v)) return false;
    *dst = (int32_t)v;
    return true;
  };
  if (!need("n_rows", &info->ny) || !need("n_columns", &info->nx) || !need("pattern_height", &info->sy) ||
      !need("pattern_width", &info->sx))
    return kpdi::fail_msg(KPDI_EINVAL, "'%s' lacks n_rows / n_columns / pattern_height / pattern_width", h.c_str());
  auto opt = [&](const char *key, double dflt) { return read_scalar(f.id, h + key, &v) ? v : dflt; };
  info->step_y = opt("step_y", 1.0);
  info->step_x = opt("step_x", 1.0);
  info->detector_pixel_size = opt("detector_pixel_size", 1.0);
  info->sample_tilt = opt("sample_tilt", 0.0);
  info->azimuth_angle = opt("azimuth_angle", 0.0);
  info->elevation_angle = opt("elevation_angle", 0.0);
  info->binning = (int32_t)opt("binning", 1.0);
  Dataset d;
  if (!open_dataset(f.id, f.scan + "/EBSD/Data/patterns", &d))
    return kpdi::fail_msg(KPDI_EINVAL, "Could not find patterns in the expected dataset 'EBSD/Data/patterns'");
  info->dtype = kpdi_dtype_of(d);
  info->n_stored = (int64_t)d.count();
  d.close();
  if (info->dtype < 0) return kpdi::fail_msg(KPDI_EINVAL, "patterns have an unsupported element type");
  Dataset bg;
  if (open_dataset(f.id, h + "static_background", &bg)) {
    info->has_static_background = bg.count() == (size_t)info->sy * info->sx ? 1 : 0;
    info->static_background_dtype = kpdi_dtype_of(bg);
    bg.close();
  }
  Dataset pc;
  info->n_pc = 0;
  if (open_dataset(f.id, h + "pcx", &pc)) {
    info->n_pc = (int64_t)pc.count();
    pc.close();
  }
  return KPDI_OK;
}

int read_into(File &f, const std::string &name, size_t want_elems, void *out, size_t *got_elems, size_t *esize) {
  Dataset d;
  if (!open_dataset(f.id, name, &d)) return kpdi::fail_msg(KPDI_EINVAL, "dataset '%s' not found", name.c_str());
  const size_t n = d.count();
  int rc = KPDI_OK;
  if (n > want_elems) {
    rc = kpdi::fail_msg(KPDI_EINVAL, "dataset '%s' holds %zu values, more than the %zu expected", name.c_str(), n,
                        want_elems);
  } else if (g_h5.Dread(d.id, d.native, 0, 0, 0, out) < 0) {
    rc = kpdi::fail_msg(KPDI_EINVAL, "reading dataset '%s' failed", name.c_str());
  }
  *got_elems = n;
  *esize = d.esize;
  d.close();
  return rc;
}

}  // namespace

extern "C" {

int kpdi_h5ebsd_info_read(const char *path, const char *scan, kpdi_h5ebsd_info *info) {
  if (!info) return kpdi::fail_msg(KPDI_EINVAL, "info pointer is NULL");
  File f;
  int rc = open_scan(path, scan, &f);
  if (rc) return rc;
  rc = fill_info(f, info);
  if (rc) return rc;
  snprintf(info->scan, sizeof info->scan, "%s", f.scan.c_str());
  return KPDI_OK;
}

int kpdi_h5ebsd_read_patterns(const char *path, const char *scan, void *out, size_t out_bytes) {
  if (!out) return kpdi::fail_msg(KPDI_EINVAL, "output pointer is NULL");
  File f;
  int rc = open_scan(path, scan, &f);
  if (rc) return rc;
  kpdi_h5ebsd_info info;
  rc = fill_info(f, &info);
  if (rc) return rc;
  const size_t want = (size_t)info.ny * info.nx * info.sy * info.sx;
  size_t got = 0, es = 0;
  // the element size is known from the info; check the caller's buffer before reading
  Dataset d;
  open_dataset(f.id, f.scan + "/EBSD/Data/patterns", &d);
  es = d.esize;
  d.close();
  if (out_bytes < want * es)
    return kpdi::fail_msg(KPDI_EINVAL, "output buffer holds %zu bytes, the scan needs %zu", out_bytes, want * es);
  rc = read_into(f, f.scan + "/EBSD/Data/patterns", want, out, &got, &es);
  if (rc) return rc;
  // file shorter than the header says: zero padding (io/plugins/_h5ebsd.py:367-378)
  if (got < want) memset((char *)out + got * es, 0, (want - got) * es);
  return KPDI_OK;
}

int kpdi_h5ebsd_read_static_background(const char *path, const char *scan, void *out, size_t out_bytes) {
  if (!out) return kpdi::fail_msg(KPDI_EINVAL, "output pointer is NULL");
  File f;
  int rc = open_scan(path, scan, &f);
  if (rc) return rc;
  kpdi_h5ebsd_info info;
  rc = fill_info(f, &info);
  if (rc) return rc;
  if (!info.has_static_background) return kpdi::fail_msg(KPDI_EINVAL, "the scan has no static background of the pattern shape");
  const size_t want = (size_t)info.sy * info.sx;
  if (out_bytes < want * kpdi_dtype_size(info.static_background_dtype))
    return kpdi::fail_msg(KPDI_EINVAL, "output buffer too small for the static background");
  size_t got = 0, es = 0;
  return read_into(f, f.scan + "/EBSD/Header/static_background", want, out, &got, &es);
}

int kpdi_h5ebsd_read_pc(const char *path, const char *scan, double *out, int64_t n_pc) {
  if (!out) return kpdi::fail_msg(KPDI_EINVAL, "output pointer is NULL");
  File f;
  int rc = open_scan(path, scan, &f);
  if (rc) return rc;
  std::vector<double> comp((size_t)n_pc);
  const char *names[3] = {"pcx", "pcy", "pcz"};
  for (int a = 0; a < 3; ++a) {
    Dataset d;
    const std::string name = f.scan + "/EBSD/Header/" + names[a];
    if (!open_dataset(f.id, name, &d)) {  // header.get("pcx", 0.5)
      for (int64_t i = 0; i < n_pc; ++i) out[3 * i + a] = 0.5;
      continue;
    }
    bool ok = (int64_t)d.count() == n_pc && d.cls == H5T_FLOAT && d.esize == 8 &&
              g_h5.Dread(d.id, d.native, 0, 0, 0, comp.data()) >= 0;
    d.close();
    if (!ok) return kpdi::fail_msg(KPDI_EINVAL, "'%s' does not hold %lld float64 values", name.c_str(), (long long)n_pc);
    for (int64_t i = 0; i < n_pc; ++i) out[3 * i + a] = comp[(size_t)i];
  }
  return KPDI_OK;
}

int kpdi_set_experimental_h5ebsd(kpdi_ctx *ctx, const char *path, const char *scan, const uint8_t *nav_mask) {
  if (!ctx) return kpdi::fail_msg(KPDI_EINVAL, "ctx is NULL");
  kpdi_h5ebsd_info info;
  int rc = kpdi_h5ebsd_info_read(path, scan, &info);
  if (rc) return rc;
  const size_t bytes = (size_t)info.ny * info.nx * info.sy * info.sx * kpdi_dtype_size(info.dtype);
  void *pinned = nullptr;
  if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess)
    return kpdi::fail_msg(KPDI_ENOMEM, "cannot pin %zu bytes of host memory", bytes);
  rc = kpdi_h5ebsd_read_patterns(path, info.scan, pinned, bytes);
  if (rc == KPDI_OK) rc = kpdi_set_experimental(ctx, pinned, info.dtype, (int64_t)info.ny * info.nx, nav_mask);
  if (rc == KPDI_OK) rc = kpdi_synchronize(ctx);  // the DMA out of the pinned buffer has finished
  (void)hipHostFree(pinned);
  return rc;
}

}  // extern "C"
