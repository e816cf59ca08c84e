// pointcloud_io.cpp -- cupoch::io for point clouds: PCD (ascii / binary / binary_compressed) and PLY
// (ascii / binary), host code.  Written against the file formats themselves (PCL's PCD v0.7
// description; the PLY 1.0 header grammar) with the behaviour of the reference's readers and
// writers: which fields make points / normals / colours (file_pcd.cu:77-124), BGR-packed rgb
// (:268-280), the header the writer emits (:572-616), doubles + uchar colours in PLY
// (file_ply.cu:361-425), non-finite points removed after reading (pointcloud_io.cpp:92-95).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "cupoch/io/class_io/pointcloud_io.h"
#include "mi_icp.h"

namespace cupoch {
namespace {

void LogWarning(const char* msg) { std::fprintf(stderr, "[cupoch_amd] Warning: %s\n", msg); }   // utility/console.h: logs, continues

typedef std::vector<Eigen::Vector3f> Vec3s;

struct HostCloud {
    Vec3s points, normals, colors;
};

void Upload(const HostCloud& h, geometry::PointCloud& pc) {
    pc.Clear();
    thrust::host_vector<Eigen::Vector3f> p(h.points.begin(), h.points.end());
    pc.SetPoints(p);
    if (h.normals.size() == h.points.size() && !h.points.empty())
        pc.SetNormals(thrust::host_vector<Eigen::Vector3f>(h.normals.begin(), h.normals.end()));
    if (h.colors.size() == h.points.size() && !h.points.empty())
        pc.SetColors(thrust::host_vector<Eigen::Vector3f>(h.colors.begin(), h.colors.end()));
}

std::string LowerExtension(const std::string& filename) {
    const size_t dot = filename.find_last_of('.');
    if (dot == std::string::npos || dot + 1 >= filename.size()) return "";
    std::string e = filename.substr(dot + 1);
    std::transform(e.begin(), e.end(), e.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return e;
}

std::vector<std::string> Split(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string t;
    while (ss >> t) out.push_back(t);
    return out;
}

// one scalar of a binary record as float: (type, size) in PCD's letters
float Scalar(const unsigned char* p, char type, int size, bool swap = false) {
    unsigned char b[8];
    if (size < 1 || size > 8) return 0.0f;   // (headers are checked against this as well: ValidScalar)
    std::memcpy(b, p, (size_t)size);
    if (swap) std::reverse(b, b + size);
    switch (type) {
        case 'I':
            if (size == 1) { int8_t v; std::memcpy(&v, b, 1); return (float)v; }
            if (size == 2) { int16_t v; std::memcpy(&v, b, 2); return (float)v; }
            if (size == 4) { int32_t v; std::memcpy(&v, b, 4); return (float)v; }
            return 0.0f;
        case 'U':
            if (size == 1) { uint8_t v; std::memcpy(&v, b, 1); return (float)v; }
            if (size == 2) { uint16_t v; std::memcpy(&v, b, 2); return (float)v; }
            if (size == 4) { uint32_t v; std::memcpy(&v, b, 4); return (float)v; }
            return 0.0f;
        case 'F':
            if (size == 4) { float v; std::memcpy(&v, b, 4); return v; }
            if (size == 8) { double v; std::memcpy(&v, b, 8); return (float)v; }
            return 0.0f;
    }
    return 0.0f;
}

// the (type, size) pairs a record may hold: everything else is rejected with the header
bool ValidScalar(char type, int size) {
    if (type == 'I' || type == 'U') return size == 1 || size == 2 || size == 4;
    if (type == 'F') return size == 4 || size == 8;
    return false;
}

// bytes left in a stream from the current position (-1: not seekable)
long long BytesLeft(std::istream& in) {
    const std::streampos here = in.tellg();
    if (here == std::streampos(-1)) return -1;
    in.seekg(0, std::ios::end);
    const std::streampos end = in.tellg();
    in.seekg(here);
    if (end == std::streampos(-1) || !in) return -1;
    return (long long)(end - here);
}

constexpr long long kMaxPoints = 0x7fffff00ll;   // what the engine takes (mi_icp_set_source / set_target)

// ---------------------------------------------------------------------------- PCD
struct PcdField {
    std::string name;
    int size = 4, count = 1, offset = 0, element = 0;  // byte offset in a record, index among the scalars
    char type = 'F';
};
struct PcdHeader {
    std::vector<PcdField> fields;
    long width = 0, height = 1, points = -1;
    int record = 0, elements = 0;
    int mode = 0;  // 0 ascii, 1 binary, 2 binary_compressed
    const PcdField* find(const char* n) const {
        for (const auto& f : fields)
            if (f.name == n) return &f;
        return nullptr;
    }
};

bool ReadPcdHeader(std::istream& in, PcdHeader& h) {
    std::string line;
    bool got_data = false;
    while (std::getline(in, line)) {
        const auto st = Split(line);
        if (st.empty() || st[0][0] == '#') continue;
        const std::string& key = st[0];
        if (key == "FIELDS" || key == "COLUMNS") {
            h.fields.assign(st.size() - 1, PcdField());
            for (size_t i = 1; i < st.size(); ++i) h.fields[i - 1].name = st[i];
        } else if (key == "SIZE" || key == "TYPE" || key == "COUNT") {
            if (st.size() - 1 != h.fields.size()) return false;
            for (size_t i = 1; i < st.size(); ++i) {
                if (key == "SIZE") h.fields[i - 1].size = std::atoi(st[i].c_str());
                else if (key == "TYPE") h.fields[i - 1].type = st[i][0];
                else h.fields[i - 1].count = std::atoi(st[i].c_str());
            }
        } else if (key == "WIDTH" && st.size() > 1) {
            h.width = std::atol(st[1].c_str());
        } else if (key == "HEIGHT" && st.size() > 1) {
            h.height = std::atol(st[1].c_str());
        } else if (key == "POINTS" && st.size() > 1) {
            h.points = std::atol(st[1].c_str());
        } else if (key == "DATA") {
            h.mode = 0;
            if (st.size() > 1) {
                if (st[1].compare(0, 17, "binary_compressed") == 0) h.mode = 2;
                else if (st[1].compare(0, 6, "binary") == 0) h.mode = 1;
            }
            got_data = true;
            break;
        }
    }
    if (!got_data || h.fields.empty()) return false;
    if (h.width < 0 || h.height < 0 || h.width > kMaxPoints || h.height > kMaxPoints) return false;
    if (h.points < 0) h.points = h.width * h.height;
    if (h.points > kMaxPoints) return false;
    long long off = 0, el = 0;
    for (auto& f : h.fields) {
        // untrusted input: a size beyond 8 would overrun the record decoder's 8-byte buffer.  Only the fields the
        // reader DECODES (x y z, normal_*, rgb[a]) must be a (type, size) pair it has a case for; any other field
        // -- a lidar's `timestamp U 8`, an `intensity I 8` -- is skipped by its size alone, as the reference skips
        // it (io/file_format/file_pcd.cu: UnpackBinaryPCDElement returns 0 for sizes it does not know and
        // CheckHeader asks for x, y, z only)
        const bool decoded = f.name == "x" || f.name == "y" || f.name == "z" || f.name == "normal_x" || f.name == "normal_y" ||
                             f.name == "normal_z" || f.name == "rgb" || f.name == "rgba";
        if (f.size < 1 || f.size > 8 || (decoded && !ValidScalar(f.type, f.size)) || f.count <= 0 || f.count > 4096) return false;
        f.offset = (int)off;
        f.element = (int)el;
        off += (long long)f.size * f.count;
        el += f.count;
        if (off > (1 << 20)) return false;
    }
    h.record = (int)off;
    h.elements = (int)el;
    return h.points > 0 && h.record > 0 && h.find("x") && h.find("y") && h.find("z");
}

Eigen::Vector3f UnpackColor(const unsigned char* p, int size) {  // packed B G R [A] (file_pcd.cu:268-280)
    if (size != 4) return Eigen::Vector3f::Zero();
    return Eigen::Vector3f((float)p[2] / 255.0f, (float)p[1] / 255.0f, (float)p[0] / 255.0f);
}

bool ReadPcdData(std::istream& in, const PcdHeader& h, HostCloud& out) {
    const PcdField *fx = h.find("x"), *fy = h.find("y"), *fz = h.find("z");
    const PcdField *nx = h.find("normal_x"), *ny = h.find("normal_y"), *nz = h.find("normal_z");
    const PcdField* fc = h.find("rgb") ? h.find("rgb") : h.find("rgba");
    const bool has_n = nx && ny && nz;
    const size_t n = (size_t)h.points;
    // the point count of the header against what the file can hold (before anything is allocated for it)
    const long long left = BytesLeft(in);
    if (left >= 0) {
        if (h.mode == 1 && (unsigned long long)left < (unsigned long long)n * (unsigned long long)h.record) return false;
        if (h.mode == 0 && (unsigned long long)left < (unsigned long long)n * 2ull * (unsigned long long)h.elements - 1ull) return false;   // "v " per value
        if (h.mode == 2 && left < 8) return false;
    }
    out.points.resize(n);
    if (has_n) out.normals.resize(n);
    if (fc) out.colors.resize(n);
    if (h.mode == 0) {
        std::string line;
        for (size_t i = 0; i < n; ++i) {
            if (!std::getline(in, line)) return false;
            const auto st = Split(line);
            if ((int)st.size() < h.elements) return false;
            auto val = [&](const PcdField* f) -> float {
                const char* s = st[(size_t)f->element].c_str();
                if (f->type == 'I') return (float)std::strtol(s, nullptr, 0);
                if (f->type == 'U') return (float)std::strtoul(s, nullptr, 0);
                return (float)std::strtod(s, nullptr);
            };
            out.points[i] = Eigen::Vector3f(val(fx), val(fy), val(fz));
            if (has_n) out.normals[i] = Eigen::Vector3f(val(nx), val(ny), val(nz));
            if (fc) {
                unsigned char b[4] = {0, 0, 0, 0};
                const char* s = st[(size_t)fc->element].c_str();
                if (fc->type == 'I') { const int32_t v = (int32_t)std::strtol(s, nullptr, 0); std::memcpy(b, &v, 4); }
                else if (fc->type == 'U') { const uint32_t v = (uint32_t)std::strtoul(s, nullptr, 0); std::memcpy(b, &v, 4); }
                else { const float v = std::strtof(s, nullptr); std::memcpy(b, &v, 4); }
                out.colors[i] = UnpackColor(b, fc->size);
            }
        }
        return true;
    }
    std::vector<unsigned char> raw;
    if (h.mode == 1) {
        raw.resize(n * (size_t)h.record);
        in.read((char*)raw.data(), (std::streamsize)raw.size());
        if ((size_t)in.gcount() != raw.size()) return false;
        for (size_t i = 0; i < n; ++i) {
            const unsigned char* r = raw.data() + i * (size_t)h.record;
            out.points[i] = Eigen::Vector3f(Scalar(r + fx->offset, fx->type, fx->size), Scalar(r + fy->offset, fy->type, fy->size),
                                            Scalar(r + fz->offset, fz->type, fz->size));
            if (has_n)
                out.normals[i] = Eigen::Vector3f(Scalar(r + nx->offset, nx->type, nx->size), Scalar(r + ny->offset, ny->type, ny->size),
                                                 Scalar(r + nz->offset, nz->type, nz->size));
            if (fc) out.colors[i] = UnpackColor(r + fc->offset, fc->size);
        }
        return true;
    }
    // binary_compressed: uint32 compressed size, uint32 uncompressed size, LZF stream; the payload
    // is field-major -- all x, then all y, ... (each field's count * size bytes per point)
    uint32_t csize = 0, usize = 0;
    in.read((char*)&csize, 4);
    in.read((char*)&usize, 4);
    if (!in || usize != (uint64_t)n * (uint64_t)h.record) return false;
    if (left >= 0 && (long long)csize > left - 8) return false;
    std::vector<unsigned char> comp(csize);
    in.read((char*)comp.data(), csize);
    if ((uint32_t)in.gcount() != csize) return false;
    raw.resize(usize);
    if ((uint64_t)mi_icp_lzf_decompress(comp.data(), csize, raw.data(), usize) != usize) return false;
    auto column = [&](const PcdField* f) { return raw.data() + (size_t)f->offset * n; };   // offset * n: fields before it
    for (size_t i = 0; i < n; ++i) {
        out.points[i] = Eigen::Vector3f(Scalar(column(fx) + i * fx->size * fx->count, fx->type, fx->size),
                                        Scalar(column(fy) + i * fy->size * fy->count, fy->type, fy->size),
                                        Scalar(column(fz) + i * fz->size * fz->count, fz->type, fz->size));
        if (has_n)
            out.normals[i] = Eigen::Vector3f(Scalar(column(nx) + i * nx->size * nx->count, nx->type, nx->size),
                                             Scalar(column(ny) + i * ny->size * ny->count, ny->type, ny->size),
                                             Scalar(column(nz) + i * nz->size * nz->count, nz->type, nz->size));
        if (fc) out.colors[i] = UnpackColor(column(fc) + i * fc->size * fc->count, fc->size);
    }
    return true;
}

float PackColor(const Eigen::Vector3f& c) {  // file_pcd.cu:618-627
    unsigned char b[4] = {0, 0, 0, 0};
    b[2] = (unsigned char)std::max(std::min((int)(c[0] * 255.0), 255), 0);
    b[1] = (unsigned char)std::max(std::min((int)(c[1] * 255.0), 255), 0);
    b[0] = (unsigned char)std::max(std::min((int)(c[2] * 255.0), 255), 0);
    float v;
    std::memcpy(&v, b, 4);
    return v;
}

// ---------------------------------------------------------------------------- PLY
struct PlyProp {
    std::string name, type, list_count, list_item;  // list_*: only for list properties
};
struct PlyElement {
    std::string name;
    long count = 0;
    std::vector<PlyProp> props;
};

int PlyTypeSize(const std::string& t, char* letter) {
    struct { const char* a; const char* b; int size; char l; } tab[] = {
            {"char", "int8", 1, 'I'},   {"uchar", "uint8", 1, 'U'},  {"short", "int16", 2, 'I'},  {"ushort", "uint16", 2, 'U'},
            {"int", "int32", 4, 'I'},   {"uint", "uint32", 4, 'U'},  {"float", "float32", 4, 'F'}, {"double", "float64", 8, 'F'}};
    for (const auto& e : tab)
        if (t == e.a || t == e.b) {
            if (letter) *letter = e.l;
            return e.size;
        }
    return 0;
}

bool ReadPly(const std::string& filename, HostCloud& out) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) return false;
    std::string line;
    if (!std::getline(in, line) || Split(line).empty() || Split(line)[0] != "ply") return false;
    std::string fmt;
    std::vector<PlyElement> elements;
    bool ended = false;
    while (std::getline(in, line)) {
        const auto st = Split(line);
        if (st.empty() || st[0] == "comment" || st[0] == "obj_info") continue;
        if (st[0] == "format" && st.size() > 1) fmt = st[1];
        else if (st[0] == "element" && st.size() > 2) {
            PlyElement e;
            e.name = st[1];
            e.count = std::atol(st[2].c_str());
            if (e.count < 0 || (e.name == "vertex" && e.count > kMaxPoints)) return false;
            elements.push_back(e);
        } else if (st[0] == "property" && !elements.empty()) {
            PlyProp p;
            if (st.size() >= 5 && st[1] == "list") {
                p.list_count = st[2];
                p.list_item = st[3];
                p.name = st[4];
            } else if (st.size() >= 3) {
                p.type = st[1];
                p.name = st[2];
            } else {
                return false;
            }
            elements.back().props.push_back(p);
        } else if (st[0] == "end_header") {
            ended = true;
            break;
        }
    }
    if (!ended || (fmt != "ascii" && fmt != "binary_little_endian" && fmt != "binary_big_endian")) return false;
    const bool ascii = fmt == "ascii", swap = fmt == "binary_big_endian";
    for (const auto& e : elements) {
        const bool vertex = e.name == "vertex";
        int ix[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};   // x y z nx ny nz red green blue -> property index
        const char* want[9] = {"x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"};
        for (size_t p = 0; p < e.props.size(); ++p)
            for (int w = 0; w < 9; ++w)
                if (e.props[p].name == want[w]) ix[w] = (int)p;
        if (vertex && (ix[0] < 0 || ix[1] < 0 || ix[2] < 0)) return false;
        const bool has_n = vertex && ix[3] >= 0 && ix[4] >= 0 && ix[5] >= 0;
        const bool has_c = vertex && ix[6] >= 0 && ix[7] >= 0 && ix[8] >= 0;
        if (vertex) {
            const long long left = BytesLeft(in);   // at least a byte per property and vertex must follow
            if (left >= 0 && (unsigned long long)left < (unsigned long long)e.count * (unsigned long long)std::max<size_t>(e.props.size(), 1)) return false;
            out.points.resize((size_t)e.count);
            if (has_n) out.normals.resize((size_t)e.count);
            if (has_c) out.colors.resize((size_t)e.count);
        }
        std::vector<float> vals(e.props.size());
        for (long i = 0; i < e.count; ++i) {
            if (ascii) {
                if (!std::getline(in, line)) return false;
                const auto st = Split(line);
                size_t t = 0;
                for (size_t p = 0; p < e.props.size(); ++p) {
                    if (!e.props[p].list_count.empty()) {   // count, then that many items
                        if (t >= st.size()) return false;
                        t += 1 + (size_t)std::atol(st[t].c_str());
                        continue;
                    }
                    if (t >= st.size()) return false;
                    vals[p] = (float)std::strtod(st[t++].c_str(), nullptr);
                }
            } else {
                for (size_t p = 0; p < e.props.size(); ++p) {
                    unsigned char b[8];
                    char letter = 'F';
                    if (!e.props[p].list_count.empty()) {
                        const int cs = PlyTypeSize(e.props[p].list_count, &letter);
                        if (!cs || !in.read((char*)b, cs)) return false;
                        const long items = (long)Scalar(b, letter, cs, swap);
                        const int is = PlyTypeSize(e.props[p].list_item, nullptr);
                        if (!is || items < 0) return false;
                        in.seekg((std::streamoff)items * is, std::ios::cur);
                        continue;
                    }
                    const int sz = PlyTypeSize(e.props[p].type, &letter);
                    if (!sz || !in.read((char*)b, sz)) return false;
                    vals[p] = Scalar(b, letter, sz, swap);
                }
            }
            if (!vertex) continue;
            out.points[(size_t)i] = Eigen::Vector3f(vals[(size_t)ix[0]], vals[(size_t)ix[1]], vals[(size_t)ix[2]]);
            if (has_n) out.normals[(size_t)i] = Eigen::Vector3f(vals[(size_t)ix[3]], vals[(size_t)ix[4]], vals[(size_t)ix[5]]);
            if (has_c)   // file_ply.cu:105-113: uchar colours scaled to [0, 1]
                out.colors[(size_t)i] = Eigen::Vector3f(vals[(size_t)ix[6]] / 255.0f, vals[(size_t)ix[7]] / 255.0f,
                                                        vals[(size_t)ix[8]] / 255.0f);
        }
        if (vertex) return true;   // everything the cloud needs has been read
    }
    return false;
}

void RemoveNonFinite(HostCloud& h, bool remove_nan, bool remove_inf) {   // pointcloud.cu:360-385
    if (!remove_nan && !remove_inf) return;
    const bool hn = h.normals.size() == h.points.size(), hc = h.colors.size() == h.points.size();
    size_t k = 0;
    for (size_t i = 0; i < h.points.size(); ++i) {
        const auto& p = h.points[i];
        const bool is_nan = remove_nan && (std::isnan(p[0]) || std::isnan(p[1]) || std::isnan(p[2]));
        const bool is_inf = remove_inf && (std::isinf(p[0]) || std::isinf(p[1]) || std::isinf(p[2]));
        if (is_nan || is_inf) continue;
        h.points[k] = p;
        if (hn) h.normals[k] = h.normals[i];
        if (hc) h.colors[k] = h.colors[i];
        ++k;
    }
    h.points.resize(k);
    if (hn) h.normals.resize(k);
    if (hc) h.colors.resize(k);
}

bool ReadHostUnguarded(const std::string& filename, const std::string& ext, HostCloud& h);

// (a header may promise more than memory holds: the readers return false, they do not throw)
bool ReadHost(const std::string& filename, const std::string& ext, HostCloud& h) {
    try {
        return ReadHostUnguarded(filename, ext, h);
    } catch (const std::exception& e) {
        LogWarning((std::string("Read geometry::PointCloud failed: ") + e.what()).c_str());
        h = HostCloud();
        return false;
    }
}

bool ReadHostUnguarded(const std::string& filename, const std::string& ext, HostCloud& h) {
    if (ext == "pcd") {
        std::ifstream in(filename, std::ios::binary);
        if (!in) {
            LogWarning(("Read PCD failed: unable to open file: " + filename).c_str());
            return false;
        }
        PcdHeader hd;
        if (!ReadPcdHeader(in, hd)) {
            LogWarning("Read PCD failed: unable to parse header.");
            return false;
        }
        if (!ReadPcdData(in, hd, h)) {
            LogWarning("Read PCD failed: unable to read data.");
            return false;
        }
        return true;
    }
    if (ext == "ply") {
        if (!ReadPly(filename, h)) {
            LogWarning(("Read PLY failed: unable to read file: " + filename).c_str());
            return false;
        }
        return true;
    }
    LogWarning("Read geometry::PointCloud failed: unknown file extension.");
    return false;
}

HostCloud Download(const geometry::PointCloud& pc) {
    HostCloud h;
    const auto p = pc.GetPoints();
    h.points.assign(p.begin(), p.end());
    if (pc.HasNormals()) {
        const auto n = pc.GetNormals();
        h.normals.assign(n.begin(), n.end());
    }
    if (pc.HasColors()) {
        const auto c = pc.GetColors();
        h.colors.assign(c.begin(), c.end());
    }
    return h;
}

}  // namespace

namespace io {

std::shared_ptr<geometry::PointCloud> CreatePointCloudFromFile(const std::string& filename, const std::string& format,
                                                               bool print_progress) {
    auto pc = std::make_shared<geometry::PointCloud>();
    ReadPointCloud(filename, *pc, format, true, true, print_progress);
    return pc;
}

bool ReadPointCloud(const std::string& filename, geometry::PointCloud& pointcloud, const std::string& format,
                    bool remove_nan_points, bool remove_infinite_points, bool /*print_progress*/) {
    const std::string ext = (format == "auto") ? LowerExtension(filename) : format;
    if (ext.empty()) {
        LogWarning("Read geometry::PointCloud failed: unknown file extension.");
        return false;
    }
    HostCloud h;
    const bool ok = ReadHost(filename, ext, h);
    if (ok) {
        RemoveNonFinite(h, remove_nan_points, remove_infinite_points);
        Upload(h, pointcloud);
    }
    return ok;
}

bool ReadPointCloudToHost(const std::string& filename, std::vector<Eigen::Vector3f>& points,
                          std::vector<Eigen::Vector3f>& normals, std::vector<Eigen::Vector3f>& colors,
                          const std::string& format, bool remove_nan_points, bool remove_infinite_points) {
    points.clear();
    normals.clear();
    colors.clear();
    const std::string ext = (format == "auto") ? LowerExtension(filename) : format;
    if (ext.empty()) {
        LogWarning("Read geometry::PointCloud failed: unknown file extension.");
        return false;
    }
    HostCloud h;
    if (!ReadHost(filename, ext, h)) return false;
    RemoveNonFinite(h, remove_nan_points, remove_infinite_points);
    points.swap(h.points);
    if (h.normals.size() == points.size()) normals.swap(h.normals);
    if (h.colors.size() == points.size()) colors.swap(h.colors);
    return true;
}

bool ReadPointCloudFromPCD(const std::string& filename, geometry::PointCloud& pointcloud, bool) {
    HostCloud h;
    if (!ReadHost(filename, "pcd", h)) return false;
    Upload(h, pointcloud);
    return true;
}

bool ReadPointCloudFromPLY(const std::string& filename, geometry::PointCloud& pointcloud, bool) {
    HostCloud h;
    if (!ReadHost(filename, "ply", h)) return false;
    Upload(h, pointcloud);
    return true;
}

bool WritePointCloudToPCD(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii,
                          bool compressed, bool) {
    if (!pointcloud.HasPoints()) {   // GenerateHeader (file_pcd.cu:521-524)
        LogWarning("Write PCD failed: unable to generate header.");
        return false;
    }
    const HostCloud h = Download(pointcloud);
    const bool hn = !h.normals.empty(), hc = !h.colors.empty();
    const size_t n = h.points.size();
    const int elements = 3 + (hn ? 3 : 0) + (hc ? 1 : 0);
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) {
        LogWarning(("Write PCD failed: unable to open file: " + filename).c_str());
        return false;
    }
    // the header the reference writes (file_pcd.cu:572-616)
    std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z%s%s\n", hn ? " normal_x normal_y normal_z" : "",
                 hc ? " rgb" : "");
    std::fprintf(f, "SIZE");
    for (int e = 0; e < elements; ++e) std::fprintf(f, " 4");
    std::fprintf(f, "\nTYPE");
    for (int e = 0; e < elements; ++e) std::fprintf(f, " F");
    std::fprintf(f, "\nCOUNT");
    for (int e = 0; e < elements; ++e) std::fprintf(f, " 1");
    std::fprintf(f, "\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA %s\n", n, n,
                 write_ascii ? "ascii" : (compressed ? "binary_compressed" : "binary"));
    bool ok = true;
    if (write_ascii) {
        for (size_t i = 0; i < n; ++i) {
            std::fprintf(f, "%.10g %.10g %.10g", h.points[i][0], h.points[i][1], h.points[i][2]);
            if (hn) std::fprintf(f, " %.10g %.10g %.10g", h.normals[i][0], h.normals[i][1], h.normals[i][2]);
            if (hc) std::fprintf(f, " %.10g", PackColor(h.colors[i]));
            std::fprintf(f, "\n");
        }
    } else {
        // record-major (binary) or field-major (binary_compressed) floats
        std::vector<float> buf((size_t)elements * n);
        auto at = [&](size_t i, int e) -> float& { return compressed ? buf[(size_t)e * n + i] : buf[i * (size_t)elements + (size_t)e]; };
        for (size_t i = 0; i < n; ++i) {
            for (int d = 0; d < 3; ++d) at(i, d) = h.points[i][d];
            int e = 3;
            if (hn) {
                for (int d = 0; d < 3; ++d) at(i, e + d) = h.normals[i][d];
                e += 3;
            }
            if (hc) at(i, e) = PackColor(h.colors[i]);
        }
        const size_t bytes = buf.size() * sizeof(float);
        if (!compressed) {
            ok = std::fwrite(buf.data(), 1, bytes, f) == bytes;
        } else if (bytes > 0xffffffffull) {
            // the format's two size words are 32 bits wide
            LogWarning("[WritePCDData] binary_compressed cannot hold more than 4 GiB of data.");
            ok = false;
        } else {
            std::vector<unsigned char> comp(bytes * 2 + 16);
            const int64_t clen = mi_icp_lzf_compress(buf.data(), (int64_t)bytes, comp.data(), (int64_t)comp.size());
            const uint32_t csize = (uint32_t)clen, usize = (uint32_t)bytes;
            if (clen <= 0 || clen > 0xffffffffll) {
                LogWarning("[WritePCDData] Failed to compress data.");
                ok = false;
            } else {
                ok = std::fwrite(&csize, 4, 1, f) == 1 && std::fwrite(&usize, 4, 1, f) == 1 &&
                     std::fwrite(comp.data(), 1, csize, f) == csize;
            }
        }
    }
    std::fclose(f);
    if (!ok) LogWarning("Write PCD failed: unable to write data.");
    return ok;
}

bool WritePointCloudToPLY(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii,
                          bool /*compressed*/, bool) {
    if (pointcloud.IsEmpty()) {   // file_ply.cu:355-358
        LogWarning("Write PLY failed: point cloud has 0 points.");
        return false;
    }
    const HostCloud h = Download(pointcloud);
    const bool hn = !h.normals.empty(), hc = !h.colors.empty();
    const size_t n = h.points.size();
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) {
        LogWarning(("Write PLY failed: unable to open file: " + filename).c_str());
        return false;
    }
    // doubles for coordinates and normals, uchar colours (file_ply.cu:361-384)
    std::fprintf(f, "ply\nformat %s 1.0\ncomment Created by cupoch_amd\nelement vertex %zu\n", write_ascii ? "ascii" : "binary_little_endian", n);
    std::fprintf(f, "property double x\nproperty double y\nproperty double z\n");
    if (hn) std::fprintf(f, "property double nx\nproperty double ny\nproperty double nz\n");
    if (hc) std::fprintf(f, "property uchar red\nproperty uchar green\nproperty uchar blue\n");
    std::fprintf(f, "end_header\n");
    auto u8 = [](float c) { return (unsigned char)std::max(std::min((int)(c * 255.0), 255), 0); };   // file_ply.cu:410-424
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
        if (write_ascii) {
            std::fprintf(f, "%.17g %.17g %.17g", (double)h.points[i][0], (double)h.points[i][1], (double)h.points[i][2]);
            if (hn) std::fprintf(f, " %.17g %.17g %.17g", (double)h.normals[i][0], (double)h.normals[i][1], (double)h.normals[i][2]);
            if (hc) std::fprintf(f, " %d %d %d", (int)u8(h.colors[i][0]), (int)u8(h.colors[i][1]), (int)u8(h.colors[i][2]));
            std::fprintf(f, "\n");
        } else {
            double d[6];
            int k = 0;
            for (int a = 0; a < 3; ++a) d[k++] = h.points[i][a];
            if (hn)
                for (int a = 0; a < 3; ++a) d[k++] = h.normals[i][a];
            ok = std::fwrite(d, sizeof(double), (size_t)k, f) == (size_t)k;
            if (hc) {
                const unsigned char c[3] = {u8(h.colors[i][0]), u8(h.colors[i][1]), u8(h.colors[i][2])};
                ok = ok && std::fwrite(c, 1, 3, f) == 3;
            }
        }
    }
    std::fclose(f);
    return ok;
}

bool WritePointCloud(const std::string& filename, const geometry::PointCloud& pointcloud, bool write_ascii, bool compressed,
                     bool print_progress) {
    const std::string ext = LowerExtension(filename);
    if (ext == "pcd") return WritePointCloudToPCD(filename, pointcloud, write_ascii, compressed, print_progress);
    if (ext == "ply") return WritePointCloudToPLY(filename, pointcloud, write_ascii, compressed, print_progress);
    LogWarning("Write geometry::PointCloud failed: unknown file extension.");
    return false;
}

}  // namespace io
}  // namespace cupoch
