// ply_io.cpp — binary PLY in the layout pcl::io::savePLYFileBinary gives a PointXYZRGB cloud
// (pose_functions.cpp:1628-1632): 15-byte vertices (x,y,z float32 + red,green,blue uchar) followed by one
// 84-byte `camera` element; the same header as the reference's bundled build/cloud.ply.
#include <cstdio>
#include <cstring>
#include <sstream>

#include "o3dr_host.h"

namespace o3dr_host {

bool save_ply_binary(const std::string& path, const PointCloud& cloud)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    std::ostringstream h;
    h << "ply\nformat binary_little_endian 1.0\ncomment PCL generated\n"
      << "element vertex " << cloud.points.size() << "\n"
      << "property float x\nproperty float y\nproperty float z\n"
      << "property uchar red\nproperty uchar green\nproperty uchar blue\n"
      << "element camera 1\n"
      << "property float view_px\nproperty float view_py\nproperty float view_pz\n"
      << "property float x_axisx\nproperty float x_axisy\nproperty float x_axisz\n"
      << "property float y_axisx\nproperty float y_axisy\nproperty float y_axisz\n"
      << "property float z_axisx\nproperty float z_axisy\nproperty float z_axisz\n"
      << "property float focal\nproperty float scalex\nproperty float scaley\n"
      << "property float centerx\nproperty float centery\n"
      << "property int viewportx\nproperty int viewporty\n"
      << "property float k1\nproperty float k2\nend_header\n";
    const std::string hs = h.str();
    fwrite(hs.data(), 1, hs.size(), f);
    std::vector<uint8_t> buf;
    buf.reserve(cloud.points.size() * 15);
    for (const PointXYZRGB& p : cloud.points) {
        uint8_t rec[15];
        memcpy(rec, &p.x, 4);
        memcpy(rec + 4, &p.y, 4);
        memcpy(rec + 8, &p.z, 4);
        rec[12] = (uint8_t)(p.rgba >> 16);
        rec[13] = (uint8_t)(p.rgba >> 8);
        rec[14] = (uint8_t)p.rgba;
        buf.insert(buf.end(), rec, rec + 15);
    }
    if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
    // camera: origin, identity axes, zeros, viewport = (width, height) = (n, 1), k1 = k2 = 0
    float cam[17] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0};
    int32_t vp[2] = {(int32_t)cloud.points.size(), 1};
    float k[2] = {0, 0};
    fwrite(cam, 4, 17, f);
    fwrite(vp, 4, 2, f);
    fwrite(k, 4, 2, f);
    fclose(f);
    return true;
}

bool read_ply(const std::string& path, PointCloud& cloud)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string header;
    int ch;
    while ((ch = fgetc(f)) != EOF) {
        header.push_back((char)ch);
        if (header.size() >= 11 && header.compare(header.size() - 11, 11, "end_header\n") == 0) break;
    }
    size_t n = 0;
    {
        const size_t p = header.find("element vertex ");
        if (p == std::string::npos || header.find("binary_little_endian") == std::string::npos) {
            fclose(f);
            return false;
        }
        n = (size_t)strtoull(header.c_str() + p + 15, nullptr, 10);
    }
    // only the x,y,z float + red,green,blue uchar vertex layout is read
    std::vector<uint8_t> buf(n * 15);
    const bool ok = fread(buf.data(), 15, n, f) == n;
    fclose(f);
    if (!ok) return false;
    cloud.points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* r = &buf[i * 15];
        PointXYZRGB p;
        memcpy(&p.x, r, 4);
        memcpy(&p.y, r + 4, 4);
        memcpy(&p.z, r + 8, 4);
        p.rgba = (255u << 24) | ((uint32_t)r[12] << 16) | ((uint32_t)r[13] << 8) | r[14];  // PCL's reader leaves a = 255
        cloud.points[i] = p;
    }
    return true;
}

}  // namespace o3dr_host
