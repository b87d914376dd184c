// Scene-file parser for the reference's text format, output-name rule and the
// 24-bit BMP writer: the data formats either side of the trace loop.
//
// Grammar and quirks follow Scene::parseScene (reference src/parse.cpp:92-271),
// which is an istream token loop (`while (scn >> str)`); this is an O(N) restatement
// of the same stream semantics over an in-memory buffer:
//   * a token beginning with '#' discards the rest of its line through
//     getline(trash, 256): a remainder of 256 or more characters puts the stream
//     in a failed state and silently ends parsing (:103-106);
//   * numbers are extracted from the CHARACTER stream, not from whitespace tokens
//     ("5.0" read as an int yields 5 and leaves ".0" as the next token); a failed
//     extraction zero-fills, still applies the directive, and ends parsing;
//   * unknown tokens are skipped one at a time (:102-267 fall-through);
//   * `material` is sticky state (:100, :139-149); vertices / normals are
//     indexed from 0 in order of appearance and range-checked (:205-216, :226-249);
//   * `normal` is normalised on read (:198); flat triangles get
//     norm(cross(v1-v0, v2-v0)) for all three normals (:462-469);
//   * `camera` stores pos/dir/up as given and halfAngle = ha * 0.01745329251
//     narrowed to float (:107-114); spot angles are stored raw (:160-163);
//   * `film_resolution W H` sets height = W in the reference (:119); that is
//     reproduced only under RT_PARSE_REF_COMPAT.
#include "../../include/rt_api.h"
#include "rt_device.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void rt_set_error(const std::string& s);

namespace {

struct Stream {
    const char* p;
    const char* end;
    bool failed = false;

    static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
    void skip_ws() { while (p < end && is_space(*p)) p++; }

    // `scn >> std::string`
    bool token(std::string& out) {
        if (failed) return false;
        skip_ws();
        if (p >= end) { failed = true; return false; }
        const char* s = p;
        while (p < end && !is_space(*p)) p++;
        out.assign(s, p - s);
        return true;
    }
    // `scn.getline(trash, 256)`
    void skip_line() {
        if (failed) return;
        int count = 0;
        while (p < end && *p != '\n' && count < 255) { p++; count++; }
        if (p < end && *p == '\n') { p++; return; }
        if (p >= end) { if (count == 0) failed = true; return; }
        failed = true;   // 255 characters consumed and the line goes on
    }
    // `scn >> float`: [+-] digits [. digits] [e [+-] digits], at least one mantissa digit
    float number_f() {
        if (failed) return 0.0f;
        skip_ws();
        const char* s = p;
        const char* q = p;
        if (q < end && (*q == '+' || *q == '-')) q++;
        int digits = 0;
        while (q < end && *q >= '0' && *q <= '9') { q++; digits++; }
        if (q < end && *q == '.') { q++; while (q < end && *q >= '0' && *q <= '9') { q++; digits++; } }
        if (digits == 0) { failed = true; return 0.0f; }
        if (q < end && (*q == 'e' || *q == 'E')) {
            const char* e = q + 1;
            if (e < end && (*e == '+' || *e == '-')) e++;
            if (e < end && *e >= '0' && *e <= '9') { while (e < end && *e >= '0' && *e <= '9') e++; q = e; }
            else { p = e; failed = true; return 0.0f; }   // "1e" is consumed and rejected
        }
        std::string tmp(s, q - s);
        p = q;
        // out of range: the extraction stores the largest finite value of that sign and fails (std::num_get,
        // [facet.num.get.virtuals]); a value too small for a float is stored as strtof rounds it, without failing
        const float v = std::strtof(tmp.c_str(), nullptr);
        if (v == HUGE_VALF) { failed = true; return 3.402823466e+38f; }
        if (v == -HUGE_VALF) { failed = true; return -3.402823466e+38f; }
        return v;
    }
    // `scn >> int`
    int number_i() {
        if (failed) return 0;
        skip_ws();
        const char* q = p;
        if (q < end && (*q == '+' || *q == '-')) q++;
        const char* d0 = q;
        while (q < end && *q >= '0' && *q <= '9') q++;
        if (q == d0) { failed = true; return 0; }
        std::string tmp(p, q - p);
        p = q;
        long long v = std::strtoll(tmp.c_str(), nullptr, 10);
        if (v > 2147483647LL) { failed = true; return 2147483647; }
        if (v < -2147483648LL) { failed = true; return (int)(-2147483647 - 1); }
        return (int)v;
    }
};

struct Builder {
    rt_scene_desc d{};
    std::vector<float> spheres, tri_v, tri_n, mats;
    std::vector<int32_t> sph_mat, tri_flag, tri_mat;
    std::vector<rt_light> lights;
    std::vector<F3> vpool, npool;
    std::string out_name = "raytraced.bmp";
    int cur_mat = 0;

    Builder() {
        // Scene::init (src/parse.cpp:69-90) and the parser's initial material (:100)
        d.width = 640; d.height = 480; d.spp = 1; d.max_depth = 5; d.proj = RT_PROJ_PERSP; d.seed = 1;
        d.cam_pos[0] = d.cam_pos[1] = d.cam_pos[2] = 0;
        d.cam_dir[0] = 0; d.cam_dir[1] = 0; d.cam_dir[2] = 1;
        d.cam_up[0] = 0; d.cam_up[1] = 1; d.cam_up[2] = 0;
        d.cam_half_angle = (float)(45.0 * 0.01745329251);
        for (int k = 0; k < 3; k++) { d.background[k] = 0; d.ambient[k] = 0; }
        const float def[14] = {0, 0, 0, 1, 1, 1, 0, 0, 0, 5, 0, 0, 0, 1};
        mats.insert(mats.end(), def, def + 14);
    }
    void push3(std::vector<float>& v, F3 a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }
};

// owner of the arrays a parsed rt_scene_desc points to
struct ParsedScene {
    rt_scene_desc desc;   // must stay first: rt_free_scene casts back
    Builder* b;
};

}  // namespace

extern "C" int rt_parse_scene_file(const char* path, int flags, rt_scene_desc** out, char* out_name,
                                   size_t out_name_cap) {
    if (!path || !out) return RT_ERR_INVALID;
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) { rt_set_error(std::string("File could not be opened: ") + path); return RT_ERR_IO; }
    std::string buf;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
    std::fclose(f);

    Builder* B = new Builder();
    Stream s{buf.data(), buf.data() + buf.size()};
    std::string tok;
    char msg[160];
    auto bad_index = [&](const char* what, int idx) {
        std::snprintf(msg, sizeof(msg), "Error: Specified %s (%d) in triangle (%d) does not exist", what, idx,
                      (int)B->tri_mat.size());
        rt_set_error(msg);
        delete B;
        return RT_ERR_PARSE;
    };
    while (s.token(tok)) {
        if (tok.empty() || tok[0] == '#') { s.skip_line(); continue; }
        if (tok == "camera") {
            float v[10];
            for (float& x : v) x = s.number_f();
            for (int k = 0; k < 3; k++) { B->d.cam_pos[k] = v[k]; B->d.cam_dir[k] = v[3 + k]; B->d.cam_up[k] = v[6 + k]; }
            B->d.cam_half_angle = (float)(v[9] * 0.01745329251);
        } else if (tok == "film_resolution") {
            int w = s.number_i(), h = s.number_i();
            B->d.width = w;
            B->d.height = (flags & RT_PARSE_REF_COMPAT) ? w : h;
        } else if (tok == "output_image") {
            std::string n;
            if (s.token(n)) B->out_name = n; else B->out_name.clear();
        } else if (tok == "sphere") {
            float v[4];
            for (float& x : v) x = s.number_f();
            B->spheres.insert(B->spheres.end(), v, v + 4);
            B->sph_mat.push_back(B->cur_mat);
        } else if (tok == "background") {
            for (int k = 0; k < 3; k++) B->d.background[k] = s.number_f();
        } else if (tok == "material") {
            float v[14];
            for (float& x : v) x = s.number_f();
            B->mats.insert(B->mats.end(), v, v + 14);
            B->cur_mat = (int)(B->mats.size() / 14) - 1;
        } else if (tok == "directional_light" || tok == "point_light") {
            float v[6];
            for (float& x : v) x = s.number_f();
            rt_light L{};
            bool dirl = tok == "directional_light";
            L.type = dirl ? RT_LIGHT_DIRECTIONAL : RT_LIGHT_POINT;
            for (int k = 0; k < 3; k++) L.color[k] = v[k];
            // Light(color, posDir, type) (src/parse.cpp:521-534): the unused one of
            // position / direction keeps its default
            L.pos[0] = L.pos[1] = L.pos[2] = 0; L.dir[0] = 0; L.dir[1] = 1; L.dir[2] = 0;
            for (int k = 0; k < 3; k++) (dirl ? L.dir : L.pos)[k] = v[3 + k];
            L.angle1 = (float)M_PI_4; L.angle2 = (float)M_PI_2;
            B->lights.push_back(L);
        } else if (tok == "spot_light") {
            float v[11];
            for (float& x : v) x = s.number_f();
            rt_light L{};
            L.type = RT_LIGHT_SPOT;
            for (int k = 0; k < 3; k++) { L.color[k] = v[k]; L.pos[k] = v[3 + k]; L.dir[k] = v[6 + k]; }
            L.angle1 = v[9]; L.angle2 = v[10];
            B->lights.push_back(L);
        } else if (tok == "ambient_light") {
            for (int k = 0; k < 3; k++) B->d.ambient[k] = s.number_f();
        } else if (tok == "max_depth") {
            B->d.max_depth = s.number_i();
        } else if (tok == "ray_type") {
            std::string t;
            s.token(t);
            if (t == "perspective") B->d.proj = RT_PROJ_PERSP;
            else if (t == "orthographic") B->d.proj = RT_PROJ_ORTHO;
            else {
                std::printf("Unknown Projection Type: %s\n", t.c_str());
                std::printf("Defaulting to Perspective\n");
                B->d.proj = RT_PROJ_PERSP;
            }
        } else if (tok == "vertex") {
            float x = s.number_f(), y = s.number_f(), z = s.number_f();
            B->vpool.push_back(f3(x, y, z));
        } else if (tok == "normal") {
            float x = s.number_f(), y = s.number_f(), z = s.number_f();
            B->npool.push_back(norm3(f3(x, y, z)));
        } else if (tok == "triangle" || tok == "normal_triangle") {
            bool smooth = tok == "normal_triangle";
            int v[3], n[3] = {0, 0, 0};
            for (int& x : v) x = s.number_i();
            if (smooth) for (int& x : n) x = s.number_i();
            int vn = (int)B->vpool.size(), nn = (int)B->npool.size();
            for (int k = 0; k < 3; k++) if (v[k] >= vn || v[k] < 0) return bad_index("vertex", v[k]);
            if (smooth) for (int k = 0; k < 3; k++) if (n[k] >= nn || n[k] < 0) return bad_index("normal", n[k]);
            F3 p0 = B->vpool[v[0]], p1 = B->vpool[v[1]], p2 = B->vpool[v[2]];
            B->push3(B->tri_v, p0); B->push3(B->tri_v, p1); B->push3(B->tri_v, p2);
            if (smooth) {
                for (int k = 0; k < 3; k++) B->push3(B->tri_n, B->npool[n[k]]);
            } else {
                F3 fn = norm3(cross3(p1 - p0, p2 - p0));
                for (int k = 0; k < 3; k++) B->push3(B->tri_n, fn);
            }
            B->tri_flag.push_back(smooth ? 1 : 0);
            B->tri_mat.push_back(B->cur_mat);
        } else if (tok == "bvh_threshold") {
            B->d.bvh_threshold = s.number_i();   // src/parse.cpp:252-256; the AUTO switch between brute force and the LBVH
        } else if (tok == "bvh_depth") {
            B->d.bvh_depth = s.number_i();       // src/parse.cpp:257-261; carried, unused (no depth cap here)
        } else if (tok == "sample_rate") {
            B->d.spp = s.number_i();
        }
        // anything else (max_vertices, max_normals, plane, rectangle, stray numbers): skipped
    }
    rt_scene_desc& d = B->d;
    d.n_spheres = (int32_t)B->sph_mat.size();
    d.n_triangles = (int32_t)B->tri_mat.size();
    d.n_materials = (int32_t)(B->mats.size() / 14);
    d.n_lights = (int32_t)B->lights.size();
    d.spheres = B->spheres.data(); d.sphere_mat = B->sph_mat.data();
    d.tri_verts = B->tri_v.data(); d.tri_normals = B->tri_n.data();
    d.tri_flags = B->tri_flag.data(); d.tri_mat = B->tri_mat.data();
    d.materials = B->mats.data(); d.lights = B->lights.data();
    ParsedScene* ps = new ParsedScene{d, B};
    *out = &ps->desc;
    if (out_name && out_name_cap) {
        std::strncpy(out_name, B->out_name.c_str(), out_name_cap - 1);
        out_name[out_name_cap - 1] = 0;
    }
    return RT_OK;
}

extern "C" void rt_free_scene(rt_scene_desc* scene) {
    if (!scene) return;
    ParsedScene* ps = reinterpret_cast<ParsedScene*>(scene);
    delete ps->b;
    delete ps;
}

// src/main.cpp:65: filename.substr(0, filename.find(".bmp") + 4).  When ".bmp" is
// absent find() returns npos and npos + 4 wraps to 3: the first three characters.
extern "C" int rt_output_name(const char* image_name, char* out, size_t cap) {
    if (!image_name || !out || cap == 0) return RT_ERR_INVALID;
    std::string n(image_name);
    size_t pos = n.find(".bmp");
    size_t len = (pos == std::string::npos) ? 3 : pos + 4;
    std::string r = n.substr(0, len);
    if (r.size() + 1 > cap) return RT_ERR_INVALID;
    std::memcpy(out, r.c_str(), r.size() + 1);
    return RT_OK;
}

// Image::Write -> BMP::SetBitDepth(24) -> BMP::WriteToFile (src/image.cpp:109-124,
// src/EasyBMP.cpp:422-598): 14-byte file header, 40-byte info header, 3780
// pels/metre both ways, rows bottom-up, B G R per pixel, rows zero-padded to 4 bytes.
extern "C" int rt_write_bmp24(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) return RT_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) { rt_set_error(std::string("cannot open for output: ") + path); return RT_ERR_IO; }
    uint32_t row = ((uint32_t)width * 3u + 3u) & ~3u;
    uint32_t pix_bytes = row * (uint32_t)height;
    uint8_t hdr[54];
    std::memset(hdr, 0, sizeof(hdr));
    auto le16 = [&](int at, uint32_t v) { hdr[at] = v & 255; hdr[at + 1] = (v >> 8) & 255; };
    auto le32 = [&](int at, uint32_t v) { le16(at, v & 0xFFFF); le16(at + 2, v >> 16); };
    hdr[0] = 'B'; hdr[1] = 'M';
    le32(2, 54 + pix_bytes);
    le32(10, 54);
    le32(14, 40);
    le32(18, (uint32_t)width);
    le32(22, (uint32_t)height);
    le16(26, 1);
    le16(28, 24);
    le32(30, 0);
    le32(34, pix_bytes);
    le32(38, 3780);
    le32(42, 3780);
    bool ok = std::fwrite(hdr, 1, 54, f) == 54;
    std::vector<uint8_t> line(row, 0);
    for (int y = height - 1; ok && y >= 0; y--) {
        const uint8_t* src = rgb + 3 * (size_t)y * width;
        for (int x = 0; x < width; x++) {
            line[3 * x + 0] = src[3 * x + 2];
            line[3 * x + 1] = src[3 * x + 1];
            line[3 * x + 2] = src[3 * x + 0];
        }
        ok = std::fwrite(line.data(), 1, row, f) == row;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { rt_set_error(std::string("short write: ") + path); return RT_ERR_IO; }
    return RT_OK;
}
