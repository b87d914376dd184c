/* The C-ABI of include/rt_api.h used from plain C: what a binding in another language (cgo, JNI, ctypes)
 * does, spelled out.  scene file -> parser -> one context on device 0 -> frame -> 24-bit BMP.
 *
 *   abi_example scene.scn [out.bmp]      renders the scene as the reference's main does (src/main.cpp:24-72)
 *   abi_example --layout                 prints size and field offsets of the structs of the header, one per
 *                                        line, for a binding to compare its own declarations with
 * Built by the Makefile with a C compiler (no C++, no HIP headers): the header has to stand on its own. */
#include "rt_api.h"

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FIELD(T, f) printf(#T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T*)0)->f))

static int layout(void) {
    printf("rt_light %zu\n", sizeof(rt_light));
    FIELD(rt_light, type); FIELD(rt_light, color); FIELD(rt_light, pos); FIELD(rt_light, dir);
    FIELD(rt_light, angle1); FIELD(rt_light, angle2);
    printf("rt_scene_desc %zu\n", sizeof(rt_scene_desc));
    FIELD(rt_scene_desc, width); FIELD(rt_scene_desc, height); FIELD(rt_scene_desc, spp); FIELD(rt_scene_desc, max_depth);
    FIELD(rt_scene_desc, proj); FIELD(rt_scene_desc, seed); FIELD(rt_scene_desc, cam_pos); FIELD(rt_scene_desc, cam_dir);
    FIELD(rt_scene_desc, cam_up); FIELD(rt_scene_desc, cam_half_angle); FIELD(rt_scene_desc, background);
    FIELD(rt_scene_desc, ambient); FIELD(rt_scene_desc, n_spheres); FIELD(rt_scene_desc, n_triangles);
    FIELD(rt_scene_desc, n_materials); FIELD(rt_scene_desc, n_lights); FIELD(rt_scene_desc, spheres);
    FIELD(rt_scene_desc, sphere_mat); FIELD(rt_scene_desc, tri_verts); FIELD(rt_scene_desc, tri_normals);
    FIELD(rt_scene_desc, tri_flags); FIELD(rt_scene_desc, tri_mat); FIELD(rt_scene_desc, materials);
    FIELD(rt_scene_desc, lights); FIELD(rt_scene_desc, bvh_threshold); FIELD(rt_scene_desc, bvh_depth);
    printf("rt_options %zu\n", sizeof(rt_options));
    FIELD(rt_options, device); FIELD(rt_options, accel); FIELD(rt_options, batch_samples); FIELD(rt_options, pool_rays);
    FIELD(rt_options, lbvh_threshold); FIELD(rt_options, stream);
    printf("rt_stats %zu\n", sizeof(rt_stats));
    FIELD(rt_stats, rays_primary); FIELD(rt_stats, rays_secondary); FIELD(rt_stats, rays_shadow);
    FIELD(rt_stats, nodes_visited); FIELD(rt_stats, sphere_tests); FIELD(rt_stats, tri_tests);
    FIELD(rt_stats, lane_node_tests); FIELD(rt_stats, lane_prim_tests);
    FIELD(rt_stats, ms_render); FIELD(rt_stats, ms_extend); FIELD(rt_stats, ms_shadow); FIELD(rt_stats, ms_shade);
    FIELD(rt_stats, ms_other); FIELD(rt_stats, launches_extend); FIELD(rt_stats, launches_shadow);
    FIELD(rt_stats, batches); FIELD(rt_stats, retries); FIELD(rt_stats, accel_used); FIELD(rt_stats, generations);
    FIELD(rt_stats, ms_host_enqueue);
    printf("rt_tree_stats %zu\n", sizeof(rt_tree_stats));
    FIELD(rt_tree_stats, primitives); FIELD(rt_tree_stats, in_use); FIELD(rt_tree_stats, depth); FIELD(rt_tree_stats, reserved);
    FIELD(rt_tree_stats, node_visits); FIELD(rt_tree_stats, prim_tests);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "--layout") == 0) return layout();
    if (argc < 2) { fprintf(stderr, "usage: abi_example scene_file [out.bmp] | --layout\n"); return 2; }
    rt_scene_desc* scene = NULL;
    char image_name[1024], out_name[1024];
    int rc = rt_parse_scene_file(argv[1], RT_PARSE_REF_COMPAT, &scene, image_name, sizeof(image_name));
    if (rc != RT_OK) { fprintf(stderr, "%s: %s\n", rt_strerror(rc), rt_last_error()); return 1; }
    if (argc >= 3) { strncpy(out_name, argv[2], sizeof(out_name) - 1); out_name[sizeof(out_name) - 1] = 0; }
    else if ((rc = rt_output_name(image_name, out_name, sizeof(out_name))) != RT_OK) { rt_free_scene(scene); return 1; }
    rt_options opt;
    memset(&opt, 0, sizeof(opt));                      /* device 0, acceleration chosen by the library */
    rt_ctx* ctx = NULL;
    uint8_t* rgb = (uint8_t*)malloc((size_t)scene->width * (size_t)scene->height * 3);
    rt_stats st;
    rc = rgb ? rt_create(&ctx, &opt) : RT_ERR_NOMEM;
    if (rc == RT_OK) rc = rt_load_scene(ctx, scene);
    if (rc == RT_OK) rc = rt_render(ctx, rgb, NULL, &st);
    if (rc == RT_OK) rc = rt_write_bmp24(out_name, rgb, scene->width, scene->height);
    if (rc != RT_OK) fprintf(stderr, "%s: %s\n", rt_strerror(rc), rt_last_error());
    else printf("%d x %d, %llu rays -> %s\n", scene->width, scene->height,
                (unsigned long long)(st.rays_primary + st.rays_secondary + st.rays_shadow), out_name);
    if (ctx) rt_destroy(ctx);
    free(rgb);
    rt_free_scene(scene);
    return rc == RT_OK ? 0 : 1;
}
