// Untrusted input to io::ReadPointCloud (cupoch_amd/cpp/src/pointcloud_io.cpp): headers that lie about sizes and
// counts must make the readers return false -- no overrun, no exception, no allocation the file cannot back.
// Host code only: built with -fsanitize=address by `make -C cupoch_amd/cpp asan` and run without a GPU
// (tests/test_io_asan.py).  Nothing here reaches the device: every file is rejected before a cloud is made.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cupoch/cupoch.h"
#include "cupoch/io/class_io/pointcloud_io.h"

using namespace cupoch;

static int failures = 0;
static void expect_rejected(const std::string& path, const char* what) {
    geometry::PointCloud pc;
    bool ok = true;
    try {
        ok = io::ReadPointCloud(path, pc);
    } catch (...) {
        std::fprintf(stderr, "EXCEPTION escaped for %s\n", what);
        ++failures;
        return;
    }
    if (ok) {
        std::fprintf(stderr, "ACCEPTED: %s\n", what);
        ++failures;
    }
}

// a file the readers must TAKE (parsed on the host only: there is no device here), with its first point checked
static void expect_points(const std::string& path, const char* what, size_t n, float x0, float y0, float z0) {
    std::vector<Eigen::Vector3f> p, nr, c;
    bool ok = false;
    try {
        ok = io::ReadPointCloudToHost(path, p, nr, c);
    } catch (...) {
        std::fprintf(stderr, "EXCEPTION escaped for %s\n", what);
        ++failures;
        return;
    }
    if (!ok || p.size() != n || (n > 0 && (p[0][0] != x0 || p[0][1] != y0 || p[0][2] != z0))) {
        std::fprintf(stderr, "REJECTED or misread: %s (ok %d, %zu points)\n", what, (int)ok, p.size());
        ++failures;
    }
}

static void write_file(const std::string& path, const std::string& header, const std::vector<unsigned char>& body = {}) {
    FILE* f = std::fopen(path.c_str(), "wb");
    std::fwrite(header.data(), 1, header.size(), f);
    if (!body.empty()) std::fwrite(body.data(), 1, body.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const std::vector<unsigned char> some(4096, 0x3f);
    const char* pcd = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\n";
    // a scalar size the record decoder has no case for (it copies `size` bytes into an 8-byte buffer)
    write_file(dir + "/size64.pcd", std::string(pcd) + "SIZE 64 64 64\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", some);
    expect_rejected(dir + "/size64.pcd", "SIZE 64");
    write_file(dir + "/size3.pcd", std::string(pcd) + "SIZE 3 3 3\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", some);
    expect_rejected(dir + "/size3.pcd", "SIZE 3 for a float");
    write_file(dir + "/type.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE Q F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", some);
    expect_rejected(dir + "/type.pcd", "unknown TYPE");
    {   // fields the reader does not decode are skipped by their size, whatever it is (a lidar's `timestamp U 8`
        // used to make the whole file unreadable; the reference reads it -- file_pcd.cu UnpackBinaryPCDElement / CheckHeader)
        std::vector<unsigned char> body;
        for (int i = 0; i < 3; ++i) {
            const float xyz[3] = {1.0f + (float)i, 2.0f, 3.0f};
            const uint64_t stamp = 0x1122334455667788ull + (uint64_t)i;
            const unsigned char odd[3] = {9, 9, 9};
            body.insert(body.end(), (const unsigned char*)xyz, (const unsigned char*)xyz + 12);
            body.insert(body.end(), (const unsigned char*)&stamp, (const unsigned char*)&stamp + 8);
            body.insert(body.end(), odd, odd + 3);
        }
        write_file(dir + "/aux.pcd", "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z timestamp ring\nSIZE 4 4 4 8 3\nTYPE F F F U I\nCOUNT 1 1 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n", body);
        expect_points(dir + "/aux.pcd", "auxiliary U 8 / I 3 fields next to x y z", 3, 1.0f, 2.0f, 3.0f);
        write_file(dir + "/aux_ascii.pcd", "VERSION 0.7\nFIELDS x y z timestamp\nSIZE 4 4 4 8\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n1 2 3 1234567890123\n4 5 6 1234567890124\n");
        expect_points(dir + "/aux_ascii.pcd", "auxiliary U 8 field, ascii", 2, 1.0f, 2.0f, 3.0f);
        // ... but not beyond the decoder's 8-byte buffer, and the decoded fields stay strict
        write_file(dir + "/aux9.pcd", "VERSION 0.7\nFIELDS x y z blob\nSIZE 4 4 4 9\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n", some);
        expect_rejected(dir + "/aux9.pcd", "SIZE 9 for an auxiliary field");
        write_file(dir + "/x8int.pcd", "VERSION 0.7\nFIELDS x y z\nSIZE 8 4 4\nTYPE I F F\nCOUNT 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n", some);
        expect_rejected(dir + "/x8int.pcd", "x as I 8 (no such case in the decoder)");
    }
    // counts the file cannot back
    write_file(dir + "/huge.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 999999999\nHEIGHT 1\nPOINTS 999999999\nDATA binary\n", some);
    expect_rejected(dir + "/huge.pcd", "POINTS beyond the file (binary)");
    write_file(dir + "/huge_ascii.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 999999999\nHEIGHT 1\nPOINTS 999999999\nDATA ascii\n1 2 3\n");
    expect_rejected(dir + "/huge_ascii.pcd", "POINTS beyond the file (ascii)");
    write_file(dir + "/toomany.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 99999999999999\nHEIGHT 1\nPOINTS 99999999999999\nDATA binary\n", some);
    expect_rejected(dir + "/toomany.pcd", "POINTS beyond what the engine takes");
    write_file(dir + "/negative.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH -5\nHEIGHT 1\nDATA binary\n", some);
    expect_rejected(dir + "/negative.pcd", "negative WIDTH");
    write_file(dir + "/count0.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 0 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", some);
    expect_rejected(dir + "/count0.pcd", "COUNT 0");
    write_file(dir + "/countbig.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 2000000000\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", some);
    expect_rejected(dir + "/countbig.pcd", "COUNT overflowing the record");
    {   // binary_compressed: a compressed size beyond the file, and sizes that do not fit the header
        std::vector<unsigned char> body(8 + 64, 0);
        const uint32_t csize = 0x7fffffffu, usize = 4 * 12;
        std::memcpy(body.data(), &csize, 4);
        std::memcpy(body.data() + 4, &usize, 4);
        write_file(dir + "/comp_big.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary_compressed\n", body);
        expect_rejected(dir + "/comp_big.pcd", "compressed size beyond the file");
        const uint32_t c2 = 64, u2 = 0xfffffff0u;
        std::memcpy(body.data(), &c2, 4);
        std::memcpy(body.data() + 4, &u2, 4);
        write_file(dir + "/comp_mismatch.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary_compressed\n", body);
        expect_rejected(dir + "/comp_mismatch.pcd", "uncompressed size that does not match the header");
        std::vector<unsigned char> garbage(8 + 48, 0xff);   // a stream that runs past its input / output
        const uint32_t c3 = 48, u3 = 48;
        std::memcpy(garbage.data(), &c3, 4);
        std::memcpy(garbage.data() + 4, &u3, 4);
        write_file(dir + "/comp_garbage.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary_compressed\n", garbage);
        expect_rejected(dir + "/comp_garbage.pcd", "LZF stream of back-references into nothing");
    }
    write_file(dir + "/nodata.pcd", std::string(pcd) + "SIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\n");
    expect_rejected(dir + "/nodata.pcd", "header without DATA");
    write_file(dir + "/mismatch.pcd", std::string(pcd) + "SIZE 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA ascii\n");
    expect_rejected(dir + "/mismatch.pcd", "SIZE with fewer entries than FIELDS");
    // PLY
    write_file(dir + "/neg.ply", "ply\nformat binary_little_endian 1.0\nelement vertex -5\nproperty float x\nproperty float y\nproperty float z\nend_header\n", some);
    expect_rejected(dir + "/neg.ply", "negative vertex count");
    write_file(dir + "/huge.ply", "ply\nformat binary_little_endian 1.0\nelement vertex 2000000000\nproperty float x\nproperty float y\nproperty float z\nend_header\n", some);
    expect_rejected(dir + "/huge.ply", "vertex count beyond the file");
    write_file(dir + "/toomany.ply", "ply\nformat ascii 1.0\nelement vertex 99999999999999\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n");
    expect_rejected(dir + "/toomany.ply", "vertex count beyond what the engine takes");
    write_file(dir + "/type.ply", "ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty quad x\nproperty float y\nproperty float z\nend_header\n", some);
    expect_rejected(dir + "/type.ply", "unknown property type");
    {   // a face element before the vertices whose list count is negative
        std::vector<unsigned char> body(64, 0);
        const int32_t items = -100;
        std::memcpy(body.data(), &items, 4);
        write_file(dir + "/list.ply", "ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list int int vertex_index\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n", body);
        expect_rejected(dir + "/list.ply", "negative list length");
    }
    write_file(dir + "/noend.ply", "ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\n");
    expect_rejected(dir + "/noend.ply", "header without end_header");
    write_file(dir + "/noxyz.ply", "ply\nformat ascii 1.0\nelement vertex 1\nproperty float a\nend_header\n1\n");
    expect_rejected(dir + "/noxyz.ply", "vertex element without x y z");
    std::printf(failures ? "FAILED %d\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
