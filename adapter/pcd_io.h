// pcd_io.h -- the prior-map file of localization mode, host only.
//   utils::readPointCloud            src/utils/superodom_utils.cpp:16-33   (pcl::PCDReader::read into a PointCloud<PointXYZI>)
//   laserMapping::initializationParam  src/LaserMapping/laserMapping.cpp:163-173  (read -> addSurfPointCloud -> overall_map message;
//                                     a file that cannot be read switches the node to mapping mode)
// (paths relative to /root/reference/super_odometry/).  PCL is not in this image and its reader is not under /root/reference: this
// restates the published PCD v0.7 file format (header VERSION / FIELDS / SIZE / TYPE / COUNT / WIDTH / HEIGHT / VIEWPOINT / POINTS /
// DATA; bodies ascii, binary = array of structures, binary_compressed = two uint32 sizes + an LZF stream of the structure of arrays)
// as far as the node needs it: the fields x, y, z (and intensity when present) of any numeric type, every other field skipped.
// Like pcl::PCDReader the points are taken as they are (NaN coordinates stay: the map insert's VoxelGrid drops non-finite points).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace so_pcd {

struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };
struct Header {
  std::vector<Field> fields;
  size_t width = 0, height = 1, points = 0, point_size = 0, data_offset = 0;
  std::string data;  // ascii | binary | binary_compressed
};

// a header number: decimal digits only, bounded -- a malformed or hostile file must end in `false` (the node then switches to
// mapping mode, laserMapping.cpp:165-171), never in an exception out of loadPriorMap()
inline bool parse_count(const std::string& tok, size_t max, size_t& out) {
  if (tok.empty() || tok.size() > 19) return false;
  size_t v = 0;
  for (char ch : tok) {
    if (ch < '0' || ch > '9') return false;
    v = v * 10 + (size_t)(ch - '0');
  }
  if (v > max) return false;
  out = v;
  return true;
}
constexpr size_t kMaxPoints = (size_t)1 << 32;  // (a prior map of 4 G points is 64 GB of xyzi: beyond anything the node could hold)

inline bool parse_header(const std::vector<uint8_t>& buf, Header& h, std::string& err) {
  size_t pos = 0;
  bool have_points = false;
  std::vector<std::string> names, sizes, types, counts;
  while (pos < buf.size()) {
    size_t eol = pos;
    while (eol < buf.size() && buf[eol] != '\n') ++eol;
    std::string line(reinterpret_cast<const char*>(buf.data()) + pos, eol - pos);
    pos = eol < buf.size() ? eol + 1 : eol;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string key;
    is >> key;
    std::vector<std::string> v;
    for (std::string t; is >> t;) v.push_back(t);
    if (key == "VERSION" || key == "VIEWPOINT") continue;
    if (key == "FIELDS" || key == "COLUMNS") names = v;
    else if (key == "SIZE") sizes = v;
    else if (key == "TYPE") types = v;
    else if (key == "COUNT") counts = v;
    else if ((key == "WIDTH" || key == "HEIGHT" || key == "POINTS") && !v.empty()) {
      size_t val = 0;
      if (!parse_count(v[0], kMaxPoints, val)) { err = "PCD header: bad " + key + " '" + v[0] + "'"; return false; }
      if (key == "WIDTH") h.width = val; else if (key == "HEIGHT") h.height = val; else { h.points = val; have_points = true; }
    }
    else if (key == "DATA" && !v.empty()) { h.data = v[0]; h.data_offset = pos; break; }
    else { err = "PCD header: unknown entry '" + key + "'"; return false; }
  }
  if (h.data.empty()) { err = "PCD header: no DATA entry"; return false; }
  if (names.empty() || sizes.size() != names.size() || types.size() != names.size()) { err = "PCD header: FIELDS / SIZE / TYPE do not match"; return false; }
  if (!counts.empty() && counts.size() != names.size()) { err = "PCD header: COUNT does not match FIELDS"; return false; }
  size_t off = 0;
  for (size_t i = 0; i < names.size(); ++i) {
    Field f;
    size_t fsize = 0, fcount = 1;
    if (!parse_count(sizes[i], 8, fsize) || (!counts.empty() && !parse_count(counts[i], 1u << 20, fcount))) {
      err = "PCD header: bad SIZE / COUNT of field '" + names[i] + "'"; return false;
    }
    f.name = names[i]; f.size = (int)fsize; f.type = types[i].empty() ? 'F' : types[i][0]; f.count = (int)fcount;
    if (!(f.size == 1 || f.size == 2 || f.size == 4 || f.size == 8) || f.count < 0 || !(f.type == 'F' || f.type == 'I' || f.type == 'U')) {
      err = "PCD header: unsupported SIZE / TYPE / COUNT of field '" + f.name + "'"; return false;
    }
    f.offset = off;
    off += (size_t)f.size * (size_t)f.count;
    h.fields.push_back(f);
  }
  h.point_size = off;
  if (off == 0 || off > ((size_t)1 << 30)) { err = "PCD header: point size out of range"; return false; }
  if (!have_points) {
    if (h.height != 0 && h.width > kMaxPoints / h.height) { err = "PCD header: WIDTH x HEIGHT out of range"; return false; }
    h.points = h.width * h.height;
  }
  if (h.points > kMaxPoints) { err = "PCD header: POINTS out of range"; return false; }
  return true;
}

inline double read_scalar(const uint8_t* p, const Field& f) {
  switch (f.type) {
    case 'F': if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } if (f.size == 8) { double v; std::memcpy(&v, p, 8); return v; } break;
    case 'I': if (f.size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; } if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; } if (f.size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; } break;
    case 'U': if (f.size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; } if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
              if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; } if (f.size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double)v; } break;
  }
  return std::nan("");
}

// LZF (Marc Lehmann's liblzf, the codec of DATA binary_compressed): control byte c < 32: c + 1 literal bytes follow; otherwise a
// back reference of length (c >> 5) + 2 (a length field of 7 is extended by the next byte) at distance ((c & 31) << 8 | next) + 1
inline bool lzf_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      const size_t n = ctrl + 1;
      if (ip + n > in_len || op + n > out_len) return false;
      std::memcpy(out + op, in + ip, n);
      ip += n; op += n;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
      if (ip >= in_len) return false;
      const size_t dist = (((size_t)ctrl & 31u) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return false;
      for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];  // (may overlap: byte by byte)
    }
  }
  return op == out_len;
}

// Reads the file into packed {x, y, z, intensity} quadruples (intensity 0 without such a field).  false + text on any failure.
inline bool read_xyzi(const std::string& path, std::vector<float>& xyzi, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) { err = "File does not exist: " + path; return false; }  // superodom_utils.cpp:17-21
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  Header h;
  if (!parse_header(buf, h, err)) return false;
  int ix = -1, iy = -1, iz = -1, ii = -1;
  for (size_t i = 0; i < h.fields.size(); ++i) {
    const std::string& n = h.fields[i].name;
    if (n == "x") ix = (int)i; else if (n == "y") iy = (int)i; else if (n == "z") iz = (int)i; else if (n == "intensity") ii = (int)i;
  }
  if (ix < 0 || iy < 0 || iz < 0) { err = "PCD file without x / y / z fields: " + path; return false; }
  for (int k : {ix, iy, iz}) if (h.fields[k].count != 1) { err = "PCD file: x / y / z with COUNT != 1"; return false; }
  const size_t n = h.points;
  const uint8_t* body = buf.data() + h.data_offset;
  const size_t body_len = buf.size() - h.data_offset;
  // the body length bounds the number of points BEFORE anything is allocated for them (POINTS is untrusted: h.point_size * n
  // cannot overflow -- both factors are bounded by parse_header -- and an ascii point takes at least two bytes per field)
  size_t n_fields_flat = 0;
  for (const Field& fd : h.fields) n_fields_flat += (size_t)fd.count;
  if (h.data == "binary" && h.point_size * n > body_len) { err = "PCD binary body shorter than POINTS x point size"; return false; }
  if (h.data == "ascii" && n_fields_flat > 0 && n > body_len / (2 * n_fields_flat) + 1) { err = "PCD ascii body shorter than POINTS"; return false; }
  if (h.data == "binary_compressed") {
    if (body_len < 8) { err = "PCD binary_compressed body too short"; return false; }
    uint32_t usize0;
    std::memcpy(&usize0, body + 4, 4);
    if ((size_t)usize0 != h.point_size * n) { err = "PCD binary_compressed: sizes do not match the header"; return false; }
  }
  try {
    xyzi.assign(4 * n, 0.f);
  } catch (const std::exception&) { err = "PCD file: no memory for " + std::to_string(n) + " points"; return false; }
  if (h.data == "ascii") {
    std::istringstream is(std::string(reinterpret_cast<const char*>(body), body_len));
    for (size_t p = 0; p < n; ++p) {
      for (size_t fi = 0; fi < h.fields.size(); ++fi)
        for (int c = 0; c < h.fields[fi].count; ++c) {
          std::string tok;
          if (!(is >> tok)) { err = "PCD ascii body ends after " + std::to_string(p) + " of " + std::to_string(n) + " points"; return false; }
          if (c != 0) continue;
          const float v = (tok == "nan" || tok == "NaN" || tok == "-nan") ? std::nanf("") : std::strtof(tok.c_str(), nullptr);
          if ((int)fi == ix) xyzi[4 * p] = v; else if ((int)fi == iy) xyzi[4 * p + 1] = v; else if ((int)fi == iz) xyzi[4 * p + 2] = v;
          else if ((int)fi == ii) xyzi[4 * p + 3] = v;
        }
    }
    return true;
  }
  std::vector<uint8_t> soa;
  bool structure_of_arrays = false;
  if (h.data == "binary_compressed") {
    if (body_len < 8) { err = "PCD binary_compressed body too short"; return false; }
    uint32_t csize, usize;
    std::memcpy(&csize, body, 4); std::memcpy(&usize, body + 4, 4);
    if ((size_t)csize + 8 > body_len || (size_t)usize != h.point_size * n) { err = "PCD binary_compressed: sizes do not match the header"; return false; }
    try { soa.resize(usize); } catch (const std::exception&) { err = "PCD binary_compressed: no memory for the decompressed body"; return false; }
    if (!lzf_decompress(body + 8, csize, soa.data(), usize)) { err = "PCD binary_compressed: corrupt LZF stream"; return false; }
    body = soa.data();
    structure_of_arrays = true;
  } else if (h.data == "binary") {
    if (h.point_size * n > body_len) { err = "PCD binary body shorter than POINTS x point size"; return false; }
  } else { err = "PCD DATA '" + h.data + "' not supported"; return false; }
  auto at = [&](int fi, size_t p) -> const uint8_t* {
    const Field& fd = h.fields[fi];
    // binary: point after point; binary_compressed: field after field, each holding its values of all points
    return structure_of_arrays ? body + fd.offset * n + p * (size_t)fd.size * (size_t)fd.count : body + p * h.point_size + fd.offset;
  };
  for (size_t p = 0; p < n; ++p) {
    xyzi[4 * p] = (float)read_scalar(at(ix, p), h.fields[ix]);
    xyzi[4 * p + 1] = (float)read_scalar(at(iy, p), h.fields[iy]);
    xyzi[4 * p + 2] = (float)read_scalar(at(iz, p), h.fields[iz]);
    if (ii >= 0 && h.fields[ii].count >= 1) xyzi[4 * p + 3] = (float)read_scalar(at(ii, p), h.fields[ii]);
  }
  return true;
}

}  // namespace so_pcd
