// Construction from files, behind the C ABI: `Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir,
// err)` (/root/reference/src/radtran/clima_radtran.f90:98-219) for hosts that do not link the reference's own loaders.
//
// Host code only.  It follows the reference's loader, src/radtran/clima_radtran_types_create.f90 --
//   read_stellar_flux :9-78, create_RTChannel / read_wavl :226-270 / :647-687, create_OpticalProperties :272-645,
//   create_Ktable :1265-1378, read_h5_Xsection :1105-1263, create_WaterContinuum :868-1046,
//   create_RayleighXsection :1048-1088, create_PhotolysisXsection :1407-1468, create_ParticleXsection :734-866,
//   the optical-properties part of the settings file src/clima_types_create.f90:578-600, :737-1000 --
// statement for statement like clima_amd/data_loader.py (the Python mirror: tests hold the two to identical tables, bit
// for bit), with the reference's error texts.  It is a CLIENT of this library's own construction entry points
// (radtran_create_begin ... radtran_create_end): nothing here touches the device.
//
// Third-party pieces, as in the Python loader: the HDF5 C library is opened at run time (dlopen: the library has no
// link-time dependency on it; CLIMA_HDF5_LIB names it); the YAML the settings and rayleigh.yaml files use (block and flow
// maps / lists, scalars, comments) is parsed here; futils v0.1.14 `addpnt`, `inter2`, `interp_discrete_to_bins` are
// restated from their published behaviour -- parity unpinned.
#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/clima_radtran_hip.h"

namespace clima_loader {

struct Fail { std::string msg; };

constexpr double C_LIGHT = 299792458.0;   // src/clima_const.f90
constexpr double RDELTA = 1.0e-4;
const double HUGE_D = std::numeric_limits<double>::max();
const double TINY_D = std::numeric_limits<double>::min();
const double LOG10TINY = std::log10(std::sqrt(std::numeric_limits<double>::min()));   // clima_const.f90:21

// ------------------------------------------------------------------------------------------------ YAML (the subset in use)
struct YNode {
  enum Kind { Null, Scalar, Map, List } kind = Null;
  std::string s;
  std::vector<std::pair<std::string, YNode>> map;
  std::vector<YNode> list;
  const YNode *get(const std::string &k) const {
    if (kind != Map) return nullptr;
    for (auto &e : map) if (e.first == k) return &e.second;
    return nullptr;
  }
};

static std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
static std::string unquote(const std::string &t) {
  std::string s = trim(t);
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
  return s;
}

struct YParser {
  struct Line { int indent; std::string text; };
  std::vector<Line> lines;
  std::string file;

  explicit YParser(const std::string &path) : file(path) {
    std::ifstream f(path);
    if (!f) throw Fail{"Could not open \"" + path + "\""};
    std::string raw;
    while (std::getline(f, raw)) {
      // comments: a '#' at the start of the line or after white space, outside quotes
      std::string t;
      char q = 0;
      for (size_t i = 0; i < raw.size(); i++) {
        const char c = raw[i];
        if (q) { if (c == q) q = 0; }
        else if (c == '"' || c == '\'') q = c;
        else if (c == '#' && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) break;
        t.push_back(c);
      }
      if (trim(t).empty() || trim(t) == "---") continue;
      int ind = 0;
      while (ind < (int)t.size() && t[ind] == ' ') ind++;
      lines.push_back(Line{ind, trim(t)});
    }
  }

  [[noreturn]] void bad() const { throw Fail{"There is an issue with formatting in \"" + file + "\""}; }

  // flow collections may run over several lines: text is taken from line `i` on until the brackets balance
  std::string flow_text(size_t &i, const std::string &first) {
    std::string t = first;
    auto depth = [](const std::string &s) {
      int d = 0; char q = 0;
      for (char c : s) {
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') q = c;
        else if (c == '{' || c == '[') d++;
        else if (c == '}' || c == ']') d--;
      }
      return d;
    };
    while (depth(t) > 0) {
      if (i >= lines.size()) bad();
      t += " " + lines[i].text;
      i++;
    }
    return t;
  }

  YNode flow(const std::string &t, size_t &p) {
    auto ws = [&] { while (p < t.size() && (t[p] == ' ' || t[p] == '\t')) p++; };
    ws();
    YNode n;
    if (p >= t.size()) return n;
    if (t[p] == '{') {
      n.kind = YNode::Map;
      p++;
      for (;;) {
        ws();
        if (p >= t.size()) bad();
        if (t[p] == '}') { p++; break; }
        size_t c = p;
        char q = 0;
        while (c < t.size() && (q || t[c] != ':')) { if (q) { if (t[c] == q) q = 0; } else if (t[c] == '"' || t[c] == '\'') q = t[c]; c++; }
        if (c >= t.size()) bad();
        const std::string key = unquote(t.substr(p, c - p));
        p = c + 1;
        n.map.emplace_back(key, flow(t, p));
        ws();
        if (p < t.size() && t[p] == ',') p++;
      }
      return n;
    }
    if (t[p] == '[') {
      n.kind = YNode::List;
      p++;
      for (;;) {
        ws();
        if (p >= t.size()) bad();
        if (t[p] == ']') { p++; break; }
        n.list.push_back(flow(t, p));
        ws();
        if (p < t.size() && t[p] == ',') p++;
      }
      return n;
    }
    size_t c = p;
    char q = 0;
    while (c < t.size() && (q || (t[c] != ',' && t[c] != '}' && t[c] != ']'))) { if (q) { if (t[c] == q) q = 0; } else if (t[c] == '"' || t[c] == '\'') q = t[c]; c++; }
    n.kind = YNode::Scalar;
    n.s = unquote(t.substr(p, c - p));
    p = c;
    if (n.s.empty() || n.s == "~" || n.s == "null") n.kind = YNode::Null;
    return n;
  }

  YNode value(size_t &i, const std::string &rest, int indent) {
    const std::string v = trim(rest);
    if (v.empty()) {   // a nested block, or nothing
      if (i < lines.size() && lines[i].indent > indent) return block(i, lines[i].indent);
      if (i < lines.size() && lines[i].indent == indent && lines[i].text.rfind("- ", 0) == 0) return block(i, indent);   // a list at the key's own indent
      return YNode();
    }
    if (v[0] == '{' || v[0] == '[') {
      const std::string t = flow_text(i, v);
      size_t p = 0;
      return flow(t, p);
    }
    YNode n;
    n.kind = YNode::Scalar;
    n.s = unquote(v);
    if (n.s == "~" || n.s == "null") n.kind = YNode::Null;
    return n;
  }

  YNode block(size_t &i, int indent) {
    YNode n;
    if (i >= lines.size()) return n;
    if (lines[i].text.rfind("- ", 0) == 0 || lines[i].text == "-") {
      n.kind = YNode::List;
      while (i < lines.size() && lines[i].indent == indent && (lines[i].text.rfind("- ", 0) == 0 || lines[i].text == "-")) {
        const std::string rest = lines[i].text.size() > 1 ? lines[i].text.substr(2) : std::string();
        i++;
        const std::string r = trim(rest);
        if (!r.empty() && r[0] != '{' && r[0] != '[' && r.find(": ") != std::string::npos) {
          // "- key: value" starts a map whose other keys follow at indent + 2
          lines.insert(lines.begin() + (long)i, Line{indent + 2, r});
          n.list.push_back(block(i, indent + 2));
        } else {
          n.list.push_back(value(i, rest, indent));
        }
      }
      return n;
    }
    n.kind = YNode::Map;
    while (i < lines.size() && lines[i].indent == indent) {
      const std::string &t = lines[i].text;
      size_t c = 0;
      char q = 0;
      while (c < t.size() && (q || !(t[c] == ':' && (c + 1 == t.size() || t[c + 1] == ' ')))) { if (q) { if (t[c] == q) q = 0; } else if (t[c] == '"' || t[c] == '\'') q = t[c]; c++; }
      if (c >= t.size()) bad();
      const std::string key = unquote(t.substr(0, c)), rest = t.substr(c + 1);
      i++;
      n.map.emplace_back(key, value(i, rest, indent));
    }
    if (i < lines.size() && lines[i].indent > indent) bad();
    return n;
  }

  YNode parse() {
    size_t i = 0;
    if (lines.empty()) return YNode();
    YNode n = block(i, lines[0].indent);
    if (i != lines.size()) bad();
    return n;
  }
};

// ------------------------------------------------------------------------------------------------ HDF5, opened at run time
typedef int64_t hid_t;
struct H5 {
  void *h = nullptr;
  int (*open)() = nullptr;
  hid_t (*Fopen)(const char *, unsigned, hid_t) = nullptr;
  int (*Fclose)(hid_t) = nullptr;
  int (*Fis_hdf5)(const char *) = nullptr;
  int (*Lexists)(hid_t, const char *, hid_t) = nullptr;
  hid_t (*Dopen2)(hid_t, const char *, hid_t) = nullptr;
  int (*Dclose)(hid_t) = nullptr;
  hid_t (*Dget_space)(hid_t) = nullptr;
  hid_t (*Dget_type)(hid_t) = nullptr;
  int (*Tget_class)(hid_t) = nullptr;
  int (*Tclose)(hid_t) = nullptr;
  int (*Sclose)(hid_t) = nullptr;
  int (*Sget_ndims)(hid_t) = nullptr;
  int (*Sget_dims)(hid_t, uint64_t *, uint64_t *) = nullptr;
  int (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *) = nullptr;
  int (*Eset_auto2)(hid_t, void *, void *) = nullptr;
  hid_t native_double = -1;
};

static H5 &h5() {
  static H5 H;
  if (H.h) return H;
  const char *env = getenv("CLIMA_HDF5_LIB");
  const char *cands[] = {env ? env : "", "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "libhdf5.so", "libhdf5_serial.so",
                         "libhdf5.so.103", "libhdf5_serial.so.103", "libhdf5.so.200", "libhdf5_serial.so.200"};
  std::string last;
  for (const char *c : cands) {
    if (!c[0]) continue;
    H.h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (H.h) break;
    if (const char *e = dlerror()) last = e;
  }
  if (!H.h) throw Fail{"the HDF5 C library was not found (set CLIMA_HDF5_LIB): " + last};
  auto sym = [&](const char *n) { void *p = dlsym(H.h, n); if (!p) throw Fail{std::string("HDF5 library lacks ") + n}; return p; };
  H.open = (int (*)())sym("H5open");
  H.Fopen = (hid_t (*)(const char *, unsigned, hid_t))sym("H5Fopen");
  H.Fclose = (int (*)(hid_t))sym("H5Fclose");
  H.Fis_hdf5 = (int (*)(const char *))sym("H5Fis_hdf5");
  H.Lexists = (int (*)(hid_t, const char *, hid_t))sym("H5Lexists");
  H.Dopen2 = (hid_t (*)(hid_t, const char *, hid_t))sym("H5Dopen2");
  H.Dclose = (int (*)(hid_t))sym("H5Dclose");
  H.Dget_space = (hid_t (*)(hid_t))sym("H5Dget_space");
  H.Dget_type = (hid_t (*)(hid_t))sym("H5Dget_type");
  H.Tget_class = (int (*)(hid_t))sym("H5Tget_class");
  H.Tclose = (int (*)(hid_t))sym("H5Tclose");
  H.Sclose = (int (*)(hid_t))sym("H5Sclose");
  H.Sget_ndims = (int (*)(hid_t))sym("H5Sget_simple_extent_ndims");
  H.Sget_dims = (int (*)(hid_t, uint64_t *, uint64_t *))sym("H5Sget_simple_extent_dims");
  H.Dread = (int (*)(hid_t, hid_t, hid_t, hid_t, hid_t, void *))sym("H5Dread");
  H.Eset_auto2 = (int (*)(hid_t, void *, void *))sym("H5Eset_auto2");
  if (H.open() < 0) throw Fail{"H5open failed"};
  H.Eset_auto2(0, nullptr, nullptr);   // errors come back through return codes here
  H.native_double = *(hid_t *)sym("H5T_NATIVE_DOUBLE_g");
  return H;
}

static bool is_file(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
static bool is_hdf5(const std::string &p) { return is_file(p) && h5().Fis_hdf5(p.c_str()) > 0; }

struct H5File {
  hid_t f = -1;
  std::string path;
  explicit H5File(const std::string &p) : path(p) {
    f = h5().Fopen(p.c_str(), 0 /* H5F_ACC_RDONLY */, 0);
    if (f < 0) throw Fail{"Failed to read \"" + p + "\"."};
  }
  ~H5File() { if (f >= 0) h5().Fclose(f); }
  bool exists(const std::string &n) const { return h5().Lexists(f, n.c_str(), 0) > 0; }
  hid_t open(const std::string &n) const {
    hid_t d = h5().Dopen2(f, n.c_str(), 0);
    if (d < 0) throw Fail{path + ": dataset \"" + n + "\" does not exist"};
    return d;
  }
  std::vector<size_t> shape(const std::string &n) const {
    hid_t d = open(n), sp = h5().Dget_space(d);
    const int nd = h5().Sget_ndims(sp);
    std::vector<uint64_t> dims((size_t)std::max(nd, 1));
    if (nd > 0) h5().Sget_dims(sp, dims.data(), nullptr);
    h5().Sclose(sp);
    h5().Dclose(d);
    return std::vector<size_t>(dims.begin(), dims.begin() + std::max(nd, 0));
  }
  bool is_float(const std::string &n) const {
    hid_t d = open(n), t = h5().Dget_type(d);
    const int cls = h5().Tget_class(t);
    h5().Tclose(t);
    h5().Dclose(d);
    return cls == 1;   // H5T_FLOAT
  }
  std::vector<double> read(const std::string &n) const {   // whole dataset as float64, C order (HDF5 converts)
    size_t cnt = 1;
    for (size_t s : shape(n)) cnt *= s;
    std::vector<double> out(cnt);
    hid_t d = open(n);
    const int rc = h5().Dread(d, h5().native_double, 0, 0, 0, out.data());
    h5().Dclose(d);
    if (rc < 0) throw Fail{path + ": could not read \"" + n + "\""};
    return out;
  }
};

// check_h5_dataset, types_create.f90:1380-1405
static void check_dataset(const H5File &h, const std::string &name, size_t ndims, const std::string &prefix) {
  if (!h.exists(name)) throw Fail{prefix + ": dataset \"" + name + "\" does not exist"};
  if (h.shape(name).size() != ndims) throw Fail{prefix + ": dataset \"" + name + "\" has wrong number of dimensions"};
  if (!h.is_float(name)) throw Fail{prefix + ": dataset \"" + name + "\" has the wrong type"};
}

// ------------------------------------------------------------------------------------------------ futils restatements
typedef std::vector<double> vec;

static bool addpnt(vec &x, vec &y, double xnew, double ynew) {   // false: unsorted input or a duplicate abscissa
  for (size_t i = 1; i < x.size(); i++) if (x[i] - x[i - 1] < 0.0) return false;
  for (double v : x) if (v == xnew) return false;
  const size_t i = (size_t)(std::lower_bound(x.begin(), x.end(), xnew) - x.begin());
  x.insert(x.begin() + (long)i, xnew);
  y.insert(y.begin() + (long)i, ynew);
  return true;
}

// bin means of the linearly connected points (x, y) on the bins with edges xg; false: grids not ascending / data do not span
static bool inter2(const vec &xg, const vec &x, const vec &y, vec &out) {
  for (size_t i = 1; i < xg.size(); i++) if (xg[i] - xg[i - 1] <= 0.0) return false;
  for (size_t i = 1; i < x.size(); i++) if (x[i] - x[i - 1] < 0.0) return false;
  if (x.empty() || x.front() > xg.front() || x.back() < xg.back()) return false;
  out.assign(xg.size() - 1, 0.0);
  const size_t n = x.size();
  size_t k = 0;
  for (size_t i = 0; i + 1 < xg.size(); i++) {
    const double xgl = xg[i], xgu = xg[i + 1];
    while (k < n - 1 && x[k + 1] <= xgl) k++;
    double area = 0.0;
    size_t j = k;
    while (j < n - 1 && x[j] < xgu) {
      const double a1 = std::max(x[j], xgl), a2 = std::min(x[j + 1], xgu);
      if (x[j + 1] != x[j] && a2 > a1) {
        const double slope = (y[j + 1] - y[j]) / (x[j + 1] - x[j]);
        const double b1 = y[j] + slope * (a1 - x[j]);
        const double b2 = y[j] + slope * (a2 - x[j]);
        area += (a2 - a1) * (b2 + b1) / 2.0;
      }
      j++;
    }
    out[i] = area / (xgu - xgl);
  }
  return true;
}

// the reference's addpnt x4 + inter2 idiom (types_create.f90:54-63, :1185-1197)
static bool pad_and_bin(const vec &wavl, vec x, vec y, double pad, vec &out) {
  if (x.empty()) return false;
  if (!addpnt(x, y, x.front() * (1.0 - RDELTA), pad)) return false;
  if (!addpnt(x, y, 0.0, pad)) return false;
  if (!addpnt(x, y, x.back() * (1.0 + RDELTA), pad)) return false;
  if (!addpnt(x, y, HUGE_D, pad)) return false;
  return inter2(wavl, x, y, out);
}

static vec interp_discrete_to_bins(const vec &wavl, const vec &x, const vec &y, bool constant, double fill) {
  for (size_t i = 1; i < x.size(); i++)
    if (x[i] - x[i - 1] <= 0.0) throw Fail{"interp_discrete_to_bins: `x` must be strictly increasing"};
  vec out;
  if (constant) {
    vec xx, yy;
    if (x[0] > 0.0) { xx.push_back(std::min(0.0, x[0] - 1.0)); yy.push_back(y[0]); }
    xx.insert(xx.end(), x.begin(), x.end());
    yy.insert(yy.end(), y.begin(), y.end());
    xx.push_back(HUGE_D);
    yy.push_back(y.back());
    if (!inter2(wavl, xx, yy, out)) throw Fail{"inter2: data do not span grid"};
    return out;
  }
  if (!pad_and_bin(wavl, x, y, fill, out)) throw Fail{"interp_discrete_to_bins: interpolation failed"};
  return out;
}

static bool strictly_increasing(const vec &v) {
  for (size_t i = 1; i < v.size(); i++) if (v[i] - v[i - 1] <= 0.0) return false;
  return true;
}
static bool is_close(double a, double b, double tol) { return std::fabs(a - b) <= tol * std::max(std::fabs(a), std::fabs(b)); }

// ------------------------------------------------------------------------------------------------ the tables
struct KTab { int sp; vec weights, log10P, temp, log10k; };
struct Xs { int type, dim, sp1, sp2; vec temp, data; };
struct Part { int p_ind; vec radii, w0, qext, gt; std::string dat; };
struct Tables {
  std::vector<std::string> species, particles;
  vec wavl, ir_wavl, sol_wavl, photons_sol;
  std::vector<KTab> k;
  std::vector<Xs> xs;
  bool has_cont = false;
  int LH2O = -1;
  vec cont_temp, cont_H2O, cont_foreign;
  std::string cont_model, k_method = "RandomOverlapResortRebin";
  std::vector<Part> part;
};

static int index_of(const std::vector<std::string> &v, const std::string &s) {
  for (size_t i = 0; i < v.size(); i++) if (v[i] == s) return (int)i;
  return -1;
}

// create_Ktable, types_create.f90:1265-1378: log10k(ngauss, npress, ntemp, nwav) in Fortran = C (nwav, ntemp, npress, ngauss),
// the [bin][T][P][g] order of the device tables -- the bytes pass through
static KTab read_ktable(const std::string &fn, int sp, vec &wavl) {
  if (!is_hdf5(fn)) throw Fail{"Failed to read \"" + fn + "\"."};
  H5File h(fn);
  const char *names[5] = {"weights", "log10P", "T", "wavelengths", "log10k"};
  const size_t nd[5] = {1, 1, 1, 1, 4};
  for (int i = 0; i < 5; i++) check_dataset(h, names[i], nd[i], fn);
  KTab k;
  k.sp = sp;
  k.weights = h.read("weights"); k.log10P = h.read("log10P"); k.temp = h.read("T");
  wavl = h.read("wavelengths");
  for (double &v : wavl) v = v * 1.0e3;
  const std::vector<size_t> shp = h.shape("log10k");
  if (!(shp[0] == wavl.size() - 1 && shp[1] == k.temp.size() && shp[2] == k.log10P.size() && shp[3] == k.weights.size()))
    throw Fail{"\"log10k\" has a bad dimension in \"" + fn + "\""};
  k.log10k = h.read("log10k");
  if (!strictly_increasing(k.log10P) || !strictly_increasing(k.temp) || k.log10P.size() < 2 || k.temp.size() < 2)
    throw Fail{"Failed to initialize interpolator for \"" + fn + "\". Error code:   1"};
  return k;
}

// rows [nT][nwav_file] of log10 values -> [nw][nT] on the bins (types_create.f90:1216-1240); raw is C (nwav_file, ntemp)
static vec regrid_rows(const std::string &fn, const vec &wavl, const vec &wav_f, const vec &raw, size_t nT) {
  const size_t nw = wavl.size() - 1, nf = wav_f.size();
  vec out(nw * nT), row(nf), r;
  for (size_t i = 0; i < nT; i++) {
    for (size_t w = 0; w < nf; w++) row[w] = raw[w * nT + i];
    if (!pad_and_bin(wavl, wav_f, row, LOG10TINY, r)) throw Fail{"Problem interpolating data in \"" + trim(fn) + "\""};
    for (size_t w = 0; w < nw; w++) out[w * nT + i] = r[w];
  }
  return out;
}

// read_h5_Xsection, types_create.f90:1105-1263
static Xs read_h5_xsection(const std::string &fn, const vec &wavl, int type, int sp1, int sp2) {
  if (!is_hdf5(fn)) throw Fail{"Failed to read \"" + fn + "\"."};
  H5File h(fn);
  if (!h.exists("log10xs")) throw Fail{fn + ": dataset \"log10xs\" does not exist"};
  const int dim = (int)h.shape("log10xs").size() - 1;
  if (dim != 0 && dim != 1) throw Fail{"Issue reading " + fn};
  check_dataset(h, "wavelengths", 1, fn);
  vec wav_f = h.read("wavelengths");
  for (double &v : wav_f) v = v * 1.0e3;
  Xs x;
  x.type = type; x.dim = dim; x.sp1 = sp1; x.sp2 = sp2;
  if (dim == 0) {
    check_dataset(h, "log10xs", 1, fn);
    vec r;
    if (!pad_and_bin(wavl, wav_f, h.read("log10xs"), LOG10TINY, r)) throw Fail{"Problem interpolating data in \"" + trim(fn) + "\""};
    x.data.resize(r.size());
    for (size_t i = 0; i < r.size(); i++) x.data[i] = std::pow(10.0, r[i]);
    return x;
  }
  check_dataset(h, "T", 1, fn);
  x.temp = h.read("T");
  check_dataset(h, "log10xs", 2, fn);
  const std::vector<size_t> shp = h.shape("log10xs");   // C (nwav_file, ntemp)
  const vec raw = h.read("log10xs");
  if (shp[1] != x.temp.size()) throw Fail{"\"log10xs\" has a bad dimension in \"" + trim(fn) + "\""};
  x.data = regrid_rows(fn, wavl, wav_f, raw, x.temp.size());
  if (x.temp.size() < 2 || !strictly_increasing(x.temp)) throw Fail{"Failed to initialize interpolator for \"" + fn + "\""};
  return x;
}

// create_WaterContinuum, types_create.f90:868-1046
static void read_water_continuum(Tables &t, const std::string &model, const std::string &fn) {
  if (index_of(t.species, "H2O") < 0) throw Fail{"\"H2O\" must be a species to include the \"continuum\" opacity"};
  if (!(t.species.size() > 1)) throw Fail{"There must be more than 1 species in order to use the \"continuum\" opacity"};
  if (!is_hdf5(fn)) throw Fail{"Continuum \"" + model + "\" is not avaliable."};
  H5File h(fn);
  check_dataset(h, "wavelengths", 1, fn);
  vec wav_f = h.read("wavelengths");
  for (double &v : wav_f) v = v * 1.0e3;
  check_dataset(h, "T", 1, fn);
  t.cont_temp = h.read("T");
  const char *names[2] = {"log10xs_H2O", "log10xs_foreign"};
  for (int n = 0; n < 2; n++) {
    check_dataset(h, names[n], 2, fn);
    const std::vector<size_t> shp = h.shape(names[n]);
    if (shp[1] != t.cont_temp.size()) throw Fail{std::string("\"") + names[n] + "\" has a bad dimension in \"" + trim(fn) + "\""};
    (n == 0 ? t.cont_H2O : t.cont_foreign) = regrid_rows(fn, t.wavl, wav_f, h.read(names[n]), t.cont_temp.size());
  }
  if (t.cont_temp.size() < 2 || !strictly_increasing(t.cont_temp)) throw Fail{"Failed to initialize interpolator for \"" + fn + "\""};
  t.has_cont = true;
  t.LH2O = index_of(t.species, "H2O");
  t.cont_model = model;
}

// create_PhotolysisXsection, types_create.f90:1407-1468 (wavelengths in nm here)
static Xs read_photolysis_xsection(const std::string &fn, const std::string &sp, int sp_ind, const vec &wavl) {
  if (!is_hdf5(fn)) throw Fail{"Species \"" + sp + "\" does not have photolysis xsection data"};
  H5File h(fn);
  check_dataset(h, "wavelengths", 1, fn);
  const vec wv = h.read("wavelengths");
  check_dataset(h, "photoabsorption", 1, fn);
  vec xs = h.read("photoabsorption");
  for (double &v : xs) v = std::log10(std::max(v, TINY_D));
  const vec r = interp_discrete_to_bins(wavl, wv, xs, false, LOG10TINY);
  Xs x;
  x.type = 3; x.dim = 0; x.sp1 = sp_ind; x.sp2 = -1;
  x.data.resize(r.size());
  for (size_t i = 0; i < r.size(); i++) x.data[i] = std::pow(10.0, r[i]);
  return x;
}

// create_ParticleXsection, types_create.f90:734-866: radii um -> cm; w0, qext, g0 (nrad, nwav) in Fortran = C (nwav, nrad)
static Part read_particle_xsection(const std::string &fn, int p_ind, const std::string &dat, const vec &wavl) {
  if (!is_hdf5(fn)) throw Fail{"Was unable to open mie data file " + trim(fn)};
  H5File h(fn);
  check_dataset(h, "wavelengths", 1, fn);
  const vec wv = h.read("wavelengths");
  check_dataset(h, "radii", 1, fn);
  Part p;
  p.p_ind = p_ind; p.dat = dat;
  p.radii = h.read("radii");
  for (double &v : p.radii) v = v / 1.0e4;
  const char *names[3] = {"w0", "qext", "g0"};
  vec raw[3];
  for (int n = 0; n < 3; n++) {
    check_dataset(h, names[n], 2, fn);
    const std::vector<size_t> shp = h.shape(names[n]);
    if (!(shp[0] == wv.size() && shp[1] == p.radii.size())) throw Fail{std::string("\"") + names[n] + "\" has the wrong shape in \"" + fn + "\""};
    raw[n] = h.read(names[n]);
  }
  const size_t nw = wavl.size() - 1, nr = p.radii.size();
  for (int n = 0; n < 3; n++) {
    vec a(nw * nr), col(wv.size());
    for (size_t i = 0; i < nr; i++) {
      for (size_t w = 0; w < wv.size(); w++) col[w] = raw[n][w * nr + i];
      const vec r = interp_discrete_to_bins(wavl, wv, col, true, 0.0);
      for (size_t w = 0; w < nw; w++) a[w * nr + i] = r[w];
    }
    (n == 0 ? p.w0 : n == 1 ? p.qext : p.gt) = a;
  }
  if (nr < 2 || !strictly_increasing(p.radii)) throw Fail{"Failed to initialize interpolator for \"" + fn + "\""};
  return p;
}

// read_stellar_flux, types_create.f90:9-78: text table (one header line; nm, mW/m^2/nm) -> mW/m^2/Hz per bin
static vec read_stellar_flux(const std::string &star_file, const vec &wavl) {
  std::ifstream f(star_file);
  if (!f) throw Fail{"The input file " + star_file + " does not exist."};
  std::string line;
  std::getline(f, line);   // header
  vec x, y;
  while (std::getline(f, line)) {
    if (trim(line).empty()) continue;
    std::istringstream is(line);
    double a, b;
    if (!(is >> a >> b)) throw Fail{"Problem reading " + star_file};
    x.push_back(a);
    y.push_back(b);
  }
  vec flux;
  if (!pad_and_bin(wavl, x, y, 0.0, flux)) throw Fail{"Problem interpolating " + trim(star_file)};
  for (size_t i = 0; i < flux.size(); i++) {
    const double wavl_av = 0.5 * (wavl[i] + wavl[i + 1]);
    flux[i] = flux[i] * (((wavl_av * 1.0e-9) * wavl_av) / C_LIGHT);   // :70-76
  }
  return flux;
}

// src/clima_eqns.f90:240-246
static double rayleigh_vardavas(double A, double B, double Delta, double lam_nm) {
  return (4.577e-21 * ((6.0 + 3.0 * Delta) / (6.0 - 7.0 * Delta)) * std::pow(A * (1.0 + B / std::pow(lam_nm * 1.0e-3, 2.0)), 2.0) *
          (1.0 / std::pow(lam_nm * 1.0e-3, 4.0)));
}

// ------------------------------------------------------------------------------------------------ settings
struct ListOrBool { bool present = false, is_list = false, on = false; std::vector<std::string> list; };
struct SettingsOpacity {   // unpack_settingsopacity, src/clima_types_create.f90:799-996
  std::string k_method;
  ListOrBool k_distributions, cia, rayleigh, photolysis_xs;
  bool has_cont = false, has_particles = false;
  std::string water_continuum;
  std::vector<std::pair<std::string, std::string>> particle_xs;
};

static std::string lower(std::string s) { for (char &c : s) c = (char)tolower(c); return s; }

static ListOrBool list_or_bool(const YNode &n, const std::string &key) {
  ListOrBool r;
  r.present = true;
  if (n.kind == YNode::List) {
    r.is_list = true;
    for (auto &e : n.list) r.list.push_back(trim(e.s));
    for (auto &x : r.list) if (std::count(r.list.begin(), r.list.end(), x) > 1) throw Fail{"\"" + x + "\" is a duplicate in " + key};
    return r;
  }
  if (n.kind == YNode::Scalar) {
    const std::string v = lower(n.s);
    if (v == "on" || v == "true" || v == "yes") { r.on = true; return r; }
    if (v == "off" || v == "false" || v == "no") { r.on = false; return r; }
  }
  throw Fail{"\"" + key + "\" must be a list or a scalar."};
}

static SettingsOpacity unpack_opacity(const YNode &op, const std::string &filename) {
  const YNode *o = op.get("opacities");
  if (!o || o->kind != YNode::Map) throw Fail{filename + ": \"opacities\" is required in \"optical-properties\""};
  SettingsOpacity s;
  if (const YNode *k = o->get("k-distributions")) {
    const YNode *m = op.get("k-method");
    s.k_method = m ? trim(m->s) : std::string();
    if (s.k_method != "RandomOverlapResortRebin") throw Fail{"k-method \"" + s.k_method + "\" in \"" + filename + "\" is not an option."};
    s.k_distributions = list_or_bool(*k, "k-distributions");
  }
  if (const YNode *n = o->get("CIA")) s.cia = list_or_bool(*n, "CIA");
  if (const YNode *n = o->get("rayleigh")) s.rayleigh = list_or_bool(*n, "rayleigh");
  if (const YNode *n = o->get("photolysis-xs")) s.photolysis_xs = list_or_bool(*n, "photolysis-xs");
  if (const YNode *n = o->get("water-continuum")) { s.has_cont = true; s.water_continuum = trim(n->s); }
  if (const YNode *n = o->get("particle-xs")) {
    if (n->kind != YNode::Null) {
      s.has_particles = true;
      for (auto &it : n->list) {
        if (it.kind != YNode::Map) throw Fail{"\"particle-xs\" entries must be dictionaries."};
        const YNode *nm = it.get("name"), *dt = it.get("data");
        if (!nm || !dt) throw Fail{"\"particle-xs\" entries must be dictionaries."};
        s.particle_xs.emplace_back(trim(nm->s), trim(dt->s));
      }
      for (auto &a : s.particle_xs) {
        int c = 0;
        for (auto &b : s.particle_xs) c += a.first == b.first;
        if (c > 1) throw Fail{"\"" + a.first + "\" is a duplicate in particle-xs"};
      }
    }
  }
  return s;
}

// parse_cia_pair, types_create.f90:689-732: split at the '-' that leaves two known species
static std::pair<int, int> parse_cia_pair(const std::string &pair_in, const std::vector<std::string> &species) {
  const std::string pair = trim(pair_in);
  if (pair.size() < 2) throw Fail{"Could not parse CIA species pair \"" + pair + "\""};
  std::vector<std::pair<int, int>> matches;
  for (size_t p = 1; p + 1 < pair.size(); p++) {
    if (pair[p] != '-') continue;
    const std::string left = trim(pair.substr(0, p)), right = trim(pair.substr(p + 1));
    if (!left.empty() && !right.empty() && index_of(species, left) >= 0 && index_of(species, right) >= 0)
      matches.emplace_back(index_of(species, left), index_of(species, right));
  }
  if (matches.empty()) throw Fail{"Could not parse CIA species pair \"" + pair + "\" into two known species."};
  if (matches.size() > 1) throw Fail{"CIA species pair \"" + pair + "\" is ambiguous; matched multiple species splits."};
  return matches[0];
}

static std::string join(const std::string &a, const std::string &b) { return a.empty() || a.back() == '/' ? a + b : a + "/" + b; }

// create_OpticalProperties, types_create.f90:272-645
static void create_optical_properties(Tables &t, const std::string &datadir, const SettingsOpacity &sop) {
  // ---- k-distributions (:298-390)
  if (!sop.k_distributions.present || (!sop.k_distributions.is_list && !sop.k_distributions.on))
    throw Fail{"You must specify at least one k-distribution in the settings file."};
  std::vector<std::string> kd = sop.k_distributions.list;
  if (!sop.k_distributions.is_list) {
    for (auto &s : t.species) if (is_file(join(join(datadir, "kdistributions"), s + ".h5"))) kd.push_back(s);
    if (kd.empty()) throw Fail{"No k-distribution data was found, but at least one k-distribution is needed."};
  }
  for (size_t i = 0; i < kd.size(); i++) {
    const std::string &sp = kd[i];
    if (index_of(t.species, sp) < 0) throw Fail{"Species \"" + sp + "\" in optical property \"k-distributions\" is not in the list of species."};
    vec wavl;
    t.k.push_back(read_ktable(join(join(datadir, "kdistributions"), sp + ".h5"), index_of(t.species, sp), wavl));
    if (i == 0) {
      t.wavl = wavl;
    } else {
      bool ok = wavl.size() == t.wavl.size();
      for (size_t w = 0; ok && w < wavl.size(); w++) ok = is_close(t.wavl[w], wavl[w], 1.0e-7);
      if (!ok) throw Fail{"Species \"" + sp + "\" has wavelength bins that do not match the wavelength bins for other species"};
    }
  }
  for (size_t i = 1; i < t.k.size(); i++) {
    bool ok = t.k[i].weights.size() == t.k[0].weights.size();
    for (size_t g = 0; ok && g < t.k[0].weights.size(); g++) ok = is_close(t.k[0].weights[g], t.k[i].weights[g], 1.0e-12);
    if (!ok) throw Fail{"All k-coeff bin weights must match."};
  }
  t.k_method = sop.k_method;

  // ---- CIA (:395-471)
  std::vector<std::string> cia_list;
  if (sop.cia.present && (sop.cia.is_list || sop.cia.on)) {
    if (!sop.cia.is_list) {
      for (auto &a : t.species)
        for (auto &b : t.species)
          if (is_file(join(join(datadir, "CIA"), a + "-" + b + ".h5")) && !(sop.has_cont && (a == "H2O" || b == "H2O")))
            cia_list.push_back(a + "-" + b);
    } else {
      cia_list = sop.cia.list;
    }
    for (auto &pair : cia_list) {
      const std::pair<int, int> ij = parse_cia_pair(pair, t.species);
      t.xs.push_back(read_h5_xsection(join(join(datadir, "CIA"), pair + ".h5"), t.wavl, 0, ij.first, ij.second));
    }
  }

  // ---- Rayleigh (:476-541)
  if (sop.rayleigh.present && (sop.rayleigh.is_list || sop.rayleigh.on)) {
    const std::string fn = join(join(datadir, "rayleigh"), "rayleigh.yaml");
    YParser yp(fn);
    const YNode root = yp.parse();
    if (root.kind != YNode::Map) throw Fail{"There is an issue with formatting in \"" + fn + "\""};
    std::vector<std::string> names;
    if (!sop.rayleigh.is_list) { for (auto &e : root.map) if (index_of(t.species, e.first) >= 0) names.push_back(e.first); }
    else names = sop.rayleigh.list;
    for (auto &sp : names) {
      if (index_of(t.species, sp) < 0) throw Fail{"Species \"" + sp + "\" in optical property \"rayleigh\" is not in the list of species."};
      const YNode *e = root.get(sp), *d = e ? e->get("data") : nullptr;
      const YNode *A = d ? d->get("A") : nullptr, *B = d ? d->get("B") : nullptr, *D = d ? d->get("Delta") : nullptr;
      char *end = nullptr;
      double v[3];
      const YNode *nodes[3] = {A, B, D};
      for (int q = 0; q < 3; q++) {
        if (!nodes[q] || nodes[q]->kind != YNode::Scalar) throw Fail{fn + ": Rayleigh data for \"" + sp + "\" is missing or malformed"};
        v[q] = std::strtod(nodes[q]->s.c_str(), &end);
        if (end == nodes[q]->s.c_str() || *end) throw Fail{fn + ": Rayleigh data for \"" + sp + "\" is missing or malformed"};
      }
      Xs x;
      x.type = 1; x.dim = 0; x.sp1 = index_of(t.species, sp); x.sp2 = -1;
      x.data.resize(t.wavl.size() - 1);
      for (size_t w = 0; w + 1 < t.wavl.size(); w++) x.data[w] = rayleigh_vardavas(v[0], v[1], v[2], t.wavl[w]);   // at the lower bin edge (:1083-1085)
      t.xs.push_back(x);
    }
  }

  // ---- photolysis cross sections (:546-591)
  if (sop.photolysis_xs.present && (sop.photolysis_xs.is_list || sop.photolysis_xs.on)) {
    std::vector<std::string> names;
    if (!sop.photolysis_xs.is_list) { for (auto &s : t.species) if (is_file(join(join(datadir, "xsections"), s + ".h5"))) names.push_back(s); }
    else names = sop.photolysis_xs.list;
    for (auto &sp : names) {
      if (index_of(t.species, sp) < 0) throw Fail{"Species \"" + sp + "\" in optical property \"photolysis-xs\" is not in the list of species."};
      t.xs.push_back(read_photolysis_xsection(join(join(datadir, "xsections"), sp + ".h5"), sp, index_of(t.species, sp), t.wavl));
    }
  }

  // ---- particles (:596-614)
  for (auto &nd : sop.particle_xs) {
    if (index_of(t.particles, nd.first) < 0) throw Fail{"Species \"" + nd.first + "\" in optical property \"particle-xs\" is not in the list of particles."};
    t.part.push_back(read_particle_xsection(join(join(join(datadir, "aerosol_xsections"), nd.second), "mie_" + nd.second + ".h5"),
                                            index_of(t.particles, nd.first), nd.second, t.wavl));
  }

  // ---- water continuum (:619-642)
  if (sop.has_cont) {
    for (auto &pair : cia_list) {
      const size_t j = pair.find('-');
      if (pair.substr(0, j) == "H2O" || pair.substr(j + 1) == "H2O")
        throw Fail{"Optical property \"water-continuum\" is set, but CIA \"" + pair + "\" is also set. This is not allowed because it would double count opacity."};
    }
    read_water_continuum(t, sop.water_continuum, join(join(datadir, "water_continuum"), sop.water_continuum + ".h5"));
  }
}

// create_RTChannel, types_create.f90:226-270 -> the channel's bin edges (a contiguous slice of the opacity grid)
static vec create_rt_channel(const std::string &datadir, const char *channel, const std::string &bins_file, const vec &wavl) {
  const std::string fn = bins_file.empty() ? join(join(datadir, "kdistributions"), "bins.h5") : bins_file;
  if (!is_hdf5(fn)) throw Fail{"Failed to read \"" + fn + "\"."};
  const std::string name = std::string(channel) + "_wavl";
  vec w;
  {
    H5File h(fn);
    check_dataset(h, name, 1, fn + "/" + name);
    w = h.read(name);
  }
  for (double &v : w) v = v * 1.0e3;
  auto argmin = [&](double x) { size_t b = 0; for (size_t i = 1; i < wavl.size(); i++) if (std::fabs(x - wavl[i]) < std::fabs(x - wavl[b])) b = i; return b; };
  const size_t i1 = argmin(w.front()), i2 = argmin(w.back());
  bool ok = i2 >= i1 && w.size() == i2 - i1 + 1;
  for (size_t i = 0; ok && i < w.size(); i++) ok = is_close(w[i], wavl[i1 + i], 1.0e-7);
  if (!ok) throw Fail{"The wavelength bins \"" + trim(fn) + "\" are not compatible with the k-distribution wavelength bins."};
  return vec(wavl.begin() + (long)i1, wavl.begin() + (long)i2 + 1);
}

static Tables load_tables(const std::string &settings_file, const std::string &star_file, const std::string &datadir) {
  YParser yp(settings_file);
  const YNode root = yp.parse();
  const YNode *op = root.get("optical-properties");
  if (root.kind != YNode::Map || !op) throw Fail{settings_file + ": \"optical-properties\" is required"};
  Tables t;
  if (const YNode *sp = op->get("species")) {
    if (const YNode *g = sp->get("gases")) for (auto &e : g->list) t.species.push_back(e.s);
    if (const YNode *p = sp->get("particles")) for (auto &e : p->list) t.particles.push_back(e.s);
  }
  const SettingsOpacity sop = unpack_opacity(*op, settings_file);
  std::string bins_file;
  if (const YNode *b = op->get("wavelength-bins-file")) if (b->kind == YNode::Scalar) bins_file = b->s;
  if (t.species.empty()) throw Fail{"\"" + settings_file + "/optical-properties/species\" does not contain any gases"};
  create_optical_properties(t, datadir, sop);
  t.ir_wavl = create_rt_channel(datadir, "ir", bins_file, t.wavl);
  t.sol_wavl = create_rt_channel(datadir, "sol", bins_file, t.wavl);
  t.photons_sol = read_stellar_flux(star_file, t.sol_wavl);
  return t;
}

}  // namespace clima_loader

extern "C" {

// Everything but the upload: the tables go into the handle (state "begun") through the library's own entry points.
// Split from radtran_create_from_files so that the loader can be checked where there is no device
// (clima_test_host_tables_digest, tests/test_loader_cabi.py).
void radtran_load_from_files(void *ptr, const char *settings_file, const char *star_file, const int *nz, const char *datadir, char *err) {
  using namespace clima_loader;
  if (err) err[0] = 0;
  auto put_err = [&](const std::string &m) { if (err) { std::strncpy(err, m.c_str(), CLIMA_ERR_LEN); err[CLIMA_ERR_LEN] = 0; } };
  try {
    const Tables t = load_tables(settings_file ? settings_file : "", star_file ? star_file : "", datadir ? datadir : "");
    const int nsp = (int)t.species.size(), np = (int)t.particles.size(), nw = (int)t.wavl.size() - 1;
    radtran_create_begin(ptr, nz, &nsp, &np, &nw, t.wavl.data(), err);
    if (err && err[0]) return;
    for (auto &k : t.k) {
      const int sp = k.sp + 1, ng = (int)k.weights.size(), nP = (int)k.log10P.size(), nT = (int)k.temp.size();
      radtran_add_ktable(ptr, &sp, &ng, k.weights.data(), &nP, k.log10P.data(), &nT, k.temp.data(), k.log10k.data(), err);
      if (err && err[0]) return;
    }
    for (auto &x : t.xs) {
      const int s1 = x.sp1 + 1, s2 = x.sp2 + 1, nT = (int)x.temp.size();
      const double zero = 0.0;
      radtran_add_xsection(ptr, &x.type, &x.dim, &s1, &s2, &nT, nT ? x.temp.data() : &zero, x.data.data(), err);
      if (err && err[0]) return;
    }
    if (t.has_cont) {
      const int L = t.LH2O + 1, nT = (int)t.cont_temp.size();
      radtran_set_water_continuum(ptr, &L, &nT, t.cont_temp.data(), t.cont_H2O.data(), t.cont_foreign.data(), err);
      if (err && err[0]) return;
    }
    for (auto &p : t.part) {
      const int pi = p.p_ind + 1, nr = (int)p.radii.size();
      radtran_add_particle(ptr, &pi, &nr, p.radii.data(), p.w0.data(), p.qext.data(), p.gt.data(), err);
      if (err && err[0]) return;
    }
    const int ni = (int)t.ir_wavl.size(), ns = (int)t.sol_wavl.size(), nps = (int)t.photons_sol.size();
    radtran_set_channels(ptr, &ni, t.ir_wavl.data(), &ns, t.sol_wavl.data(), err);
    if (err && err[0]) return;
    radtran_set_photons_sol(ptr, &nps, t.photons_sol.data(), err);
    if (err && err[0]) return;
    // names for opacities2yaml (clima_radtran_types.f90:328-430)
    std::string sn, pn, dn;
    for (size_t i = 0; i < t.species.size(); i++) sn += (i ? "\n" : "") + t.species[i];
    for (size_t i = 0; i < t.particles.size(); i++) pn += (i ? "\n" : "") + t.particles[i];
    for (size_t i = 0; i < t.part.size(); i++) dn += (i ? "\n" : "") + t.part[i].dat;
    radtran_set_names(ptr, sn.c_str(), pn.c_str(), err);
    if (err && err[0]) return;
    radtran_set_opacity_labels(ptr, t.k_method.c_str(), t.has_cont ? t.cont_model.c_str() : "", dn.c_str(), err);
  } catch (const Fail &f) {
    put_err(f.msg);
  } catch (const std::exception &e) {
    put_err(e.what());
  }
}

// `Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`, src/radtran/clima_radtran.f90:98-126
void radtran_create_from_files(void *ptr, const char *settings_file, const char *star_file, const int *num_zenith_angles,
                               const double *surface_albedo, const int *nz, const char *datadir, char *err) {
  radtran_load_from_files(ptr, settings_file, star_file, nz, datadir, err);
  if (err && err[0]) return;
  radtran_create_end(ptr, num_zenith_angles, surface_albedo, err);
}

}  // extern "C"
