// cilantro_hip/point_cloud.hpp -- the wire format either side of the ICP path: a
// minimal PointCloud3f with PLY ingest / output, header-only, no third-party code.  Mirrors what the reference's
// examples do before and after registration:
//   utilities/point_cloud.hpp:20-22, :501-541     PointCloud3f {points, normals, colors}, fromPLYFile / toPLYFile
//   utilities/ply_io.hpp:43-143                   PLYReader / PLYWriter (tinyply underneath in the reference)
// Same property names as the reference reads and writes: vertex x y z [nx ny nz] [red green blue]; colours are
// uchar in the file and floats in [0,1] in memory (point_cloud.hpp:513, :535).  Reads ascii and
// binary_little_endian files with any scalar property types; other elements (faces ...) are skipped.
// The clouds feed the engine as non-owning views:  ConstPointsView(cloud.points), ConstPointsView(cloud.normals).
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cilantro_hip {

class PointCloud3f {
public:
  std::vector<float> points;   // packed xyz (3 x N column-major, as cilantro's VectorSet3f)
  std::vector<float> normals;  // packed xyz or empty
  std::vector<float> colors;   // packed rgb in [0,1] or empty

  PointCloud3f() = default;
  explicit PointCloud3f(const std::string& file_name) { fromPLYFile(file_name); }   // point_cloud.hpp:118-121

  size_t size() const { return points.size() / 3; }
  bool isEmpty() const { return points.empty(); }
  bool hasNormals() const { return size() > 0 && normals.size() == points.size(); }   // point_cloud.hpp:139-141
  bool hasColors() const { return size() > 0 && colors.size() == points.size(); }

  PointCloud3f& fromPLYFile(const std::string& file_name) {
    std::ifstream f(file_name, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + file_name);
    std::string line;
    if (!std::getline(f, line) || strip(line) != "ply") throw std::runtime_error(file_name + ": not a PLY file");
    enum Fmt { ASCII, BLE } fmt = ASCII;
    struct Prop { std::string name, type; bool is_list; std::string count_type; };
    struct Elem { std::string name; size_t count; std::vector<Prop> props; };
    std::vector<Elem> elems;
    for (;;) {
      if (!std::getline(f, line)) throw std::runtime_error(file_name + ": truncated PLY header");
      std::istringstream ss(strip(line));
      std::string key;
      ss >> key;
      if (key == "end_header") break;
      if (key == "format") {
        std::string v;
        ss >> v;
        if (v == "ascii") fmt = ASCII;
        else if (v == "binary_little_endian") fmt = BLE;
        else throw std::runtime_error(file_name + ": unsupported PLY format " + v);
      } else if (key == "element") {
        Elem e;
        ss >> e.name >> e.count;
        elems.push_back(e);
      } else if (key == "property") {
        if (elems.empty()) throw std::runtime_error(file_name + ": property before element");
        Prop p;
        std::string t;
        ss >> t;
        if (t == "list") { p.is_list = true; ss >> p.count_type >> p.type >> p.name; }
        else { p.is_list = false; p.type = t; ss >> p.name; }
        elems.back().props.push_back(p);
      }
    }
    points.clear(); normals.clear(); colors.clear();
    // The element counts are untrusted input: a count larger than the rest of the file can hold (every row takes at
    // least one byte per property) is a malformed file, not an allocation request.
    const std::streamoff data_begin = (std::streamoff)f.tellg();
    f.seekg(0, std::ios::end);
    const std::streamoff data_bytes = (std::streamoff)f.tellg() - data_begin;
    f.seekg(data_begin, std::ios::beg);
    if (!f || data_bytes < 0) throw std::runtime_error(file_name + ": cannot size the PLY data section");
    for (const Elem& e : elems) {
      size_t min_row = 0;
      for (const Prop& pr : e.props) min_row += fmt == ASCII ? 1 : scalar_size(pr.is_list ? pr.count_type : pr.type, file_name);
      if (min_row == 0) min_row = 1;
      if (e.count > (size_t)data_bytes / min_row) throw std::runtime_error(file_name + ": element count exceeds the file size");
    }
    for (const Elem& e : elems) {
      const bool vertex = e.name == "vertex";
      int ix[9];
      for (int& v : ix) v = -1;
      static const char* names[9] = {"x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"};
      for (size_t p = 0; p < e.props.size(); ++p)
        for (int k = 0; k < 9; ++k)
          if (!e.props[p].is_list && e.props[p].name == names[k]) ix[k] = (int)p;
      const bool has_p = vertex && ix[0] >= 0 && ix[1] >= 0 && ix[2] >= 0;
      const bool has_n = vertex && ix[3] >= 0 && ix[4] >= 0 && ix[5] >= 0;
      const bool has_c = vertex && ix[6] >= 0 && ix[7] >= 0 && ix[8] >= 0;
      if (has_p) points.resize(3 * e.count);
      if (has_n) normals.resize(3 * e.count);
      if (has_c) colors.resize(3 * e.count);
      std::vector<double> row(e.props.size());
      for (size_t i = 0; i < e.count; ++i) {
        for (size_t p = 0; p < e.props.size(); ++p) {
          const Prop& pr = e.props[p];
          if (pr.is_list) {
            const size_t cnt = (size_t)read_scalar(f, pr.count_type, fmt == ASCII, file_name);
            for (size_t k = 0; k < cnt; ++k) (void)read_scalar(f, pr.type, fmt == ASCII, file_name);
            row[p] = 0.0;
          } else {
            row[p] = read_scalar(f, pr.type, fmt == ASCII, file_name);
          }
        }
        if (has_p) for (int k = 0; k < 3; ++k) points[3 * i + k] = (float)row[ix[k]];
        if (has_n) for (int k = 0; k < 3; ++k) normals[3 * i + k] = (float)row[ix[3 + k]];
        if (has_c) for (int k = 0; k < 3; ++k) colors[3 * i + k] = (float)row[ix[6 + k]] * (1.0f / 255.0f);
      }
      if (vertex) break;   // everything the path needs has been read
    }
    return *this;
  }

  const PointCloud3f& toPLYFile(const std::string& file_name, bool binary = true) const {
    std::ofstream f(file_name, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + file_name + " for writing");
    const size_t n = size();
    f << "ply\nformat " << (binary ? "binary_little_endian" : "ascii") << " 1.0\nelement vertex " << n
      << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (hasNormals()) f << "property float nx\nproperty float ny\nproperty float nz\n";
    if (hasColors()) f << "property uchar red\nproperty uchar green\nproperty uchar blue\n";
    f << "end_header\n";
    for (size_t i = 0; i < n; ++i) {
      unsigned char rgb[3] = {0, 0, 0};
      if (hasColors())
        for (int k = 0; k < 3; ++k) {
          const float v = 255.0f * colors[3 * i + k];            // point_cloud.hpp:535: cast, not rounded
          rgb[k] = (unsigned char)(v < 0.0f ? 0.0f : v > 255.0f ? 255.0f : v);
        }
      if (binary) {
        f.write(reinterpret_cast<const char*>(&points[3 * i]), 12);
        if (hasNormals()) f.write(reinterpret_cast<const char*>(&normals[3 * i]), 12);
        if (hasColors()) f.write(reinterpret_cast<const char*>(rgb), 3);
      } else {
        std::ostringstream ss;
        ss.precision(9);
        ss << points[3 * i] << ' ' << points[3 * i + 1] << ' ' << points[3 * i + 2];
        if (hasNormals()) ss << ' ' << normals[3 * i] << ' ' << normals[3 * i + 1] << ' ' << normals[3 * i + 2];
        if (hasColors()) ss << ' ' << (int)rgb[0] << ' ' << (int)rgb[1] << ' ' << (int)rgb[2];
        f << ss.str() << '\n';
      }
    }
    if (!f) throw std::runtime_error("write failed: " + file_name);
    return *this;
  }

private:
  static std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
    return s.substr(a, b - a);
  }
  template <typename T>
  static double read_bin(std::istream& f, const std::string& file) {
    T v;
    f.read(reinterpret_cast<char*>(&v), sizeof(T));   // the host is little endian (x86-64 / the GPU boxes)
    if (!f) throw std::runtime_error(file + ": truncated PLY data");
    return (double)v;
  }
  static size_t scalar_size(const std::string& type, const std::string& file) {
    if (type == "double" || type == "float64") return 8;
    if (type == "float" || type == "float32" || type == "uint" || type == "uint32" || type == "int" || type == "int32") return 4;
    if (type == "ushort" || type == "uint16" || type == "short" || type == "int16") return 2;
    if (type == "uchar" || type == "uint8" || type == "char" || type == "int8") return 1;
    throw std::runtime_error(file + ": unknown PLY property type " + type);
  }
  static double read_scalar(std::istream& f, const std::string& type, bool ascii, const std::string& file) {
    if (ascii) {
      double v;
      if (!(f >> v)) throw std::runtime_error(file + ": truncated PLY data");
      return v;
    }
    if (type == "float" || type == "float32") return read_bin<float>(f, file);
    if (type == "double" || type == "float64") return read_bin<double>(f, file);
    if (type == "uchar" || type == "uint8") return read_bin<uint8_t>(f, file);
    if (type == "char" || type == "int8") return read_bin<int8_t>(f, file);
    if (type == "ushort" || type == "uint16") return read_bin<uint16_t>(f, file);
    if (type == "short" || type == "int16") return read_bin<int16_t>(f, file);
    if (type == "uint" || type == "uint32") return read_bin<uint32_t>(f, file);
    if (type == "int" || type == "int32") return read_bin<int32_t>(f, file);
    throw std::runtime_error(file + ": unknown PLY property type " + type);
  }
};

}  // namespace cilantro_hip
