/* ch_main.c -- `corona-mi`: command line renderer with the reference's interface
 * (src/main.c:415-437, usage 420-428; src/view.c:275-292; src/display.d/null.c:50-75):
 *
 *   corona-mi <scene.nra2> [-s spp] [-w width] [-h height] [-x postfix] [--frame n] [--batch n]
 *             [--sampler pt|ptdl] [--pointsampler rand|halton] [--max-verts n] [--rgb2spec lut]  [--iso v] [-c cam] [--info] [--device-build]
 *             [--gpus n | --devices i,j,...] [--traversal exact|fast] [--wavelengths 1|4]
 *
 * --gpus n renders on the first n GPUs of the node, --devices on the listed ones (a device may be named twice): every batch's path
 * indices are split over them (mi_group_render) and the framebuffers are added up on the first one (mi_group_fb_reduce: RCCL over
 * xGMI) before the image is read back -- the single host thread stands where the reference's worker pool stood.
 *
 * --wavelengths 4: hero wavelengths, four per path -- what a reference built with -DMF_COUNT=4 renders (mi_scene_set_wavelengths).
 *
 * --info validates the scene files (.nra2, .geo, .cam) on the host and prints what the backend would get, without
 * touching a GPU (validators for the on-disk formats).
 *
 * Progression loop of view_render() (src/view.c:630-695) with the pthread pool dispatch (643-645)
 * replaced by one mi_group_render() per batch; writes <basename><postfix>_fb00.pfm like view_write_images
 * (src/view.c:543-552) and a sidecar with the mean image and timings.
 * -t (threads) is accepted and ignored: the workers are the GPU's wavefronts.
 */
#include "ch_host.h"
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

static double now(void)
{
  struct timeval t; gettimeofday(&t, 0);
  return t.tv_sec + 1e-6*t.tv_usec;
}

int main(int argc, char *argv[])
{
  if(argc < 2)
  {
    fprintf(stderr, "usage: %s <scene.nra2> [-s spp] [-w w] [-h h] [-x postfix] [--frame n] [--batch n]\n"
                    "          [--sampler pt|ptdl] [--pointsampler rand|halton] [--max-verts n] [--rgb2spec ergb2spec.coeff] [--iso v] [-c file.cam] [--info]\n", argv[0]);
    return 1;
  }
  ch_options opt;
  memset(&opt, 0, sizeof(opt));
  opt.verbose = 1;
  uint64_t spp = 10, batch = 1;              /* display_open default: 10 progressions */
  const char *postfix = "render";
  int info_only = 0, device_build = 0, traversal = -1, wavelengths = 1;
  int devices[64], num_devices = 0;
  for(int i=2;i<argc;i++)
  {
    if(!strcmp(argv[i], "--gpus") && i+1 < argc)
    {
      num_devices = atoi(argv[++i]);
      if(num_devices < 1 || num_devices > 64) { fprintf(stderr, "[main] --gpus takes 1..64\n"); return 1; }
      for(int k=0;k<num_devices;k++) devices[k] = k;
      continue;
    }
    if(!strcmp(argv[i], "--devices") && i+1 < argc)
    {
      num_devices = 0;
      for(char *tok = strtok(argv[++i], ","); tok && num_devices < 64; tok = strtok(0, ",")) devices[num_devices++] = atoi(tok);
      if(!num_devices) { fprintf(stderr, "[main] --devices takes a list like 0,1,2\n"); return 1; }
      continue;
    }
    if(!strcmp(argv[i], "--traversal") && i+1 < argc)
    { /* exact | fast | auto (= the scene's default); anything else is an error, not silently FAST */
      const char *t = argv[++i];
      if(!strcmp(t, "exact")) traversal = MI_TRAVERSAL_EXACT;
      else if(!strcmp(t, "fast")) traversal = MI_TRAVERSAL_FAST;
      else if(!strcmp(t, "auto")) traversal = -1;
      else { fprintf(stderr, "[corona-mi] --traversal %s: expected exact, fast or auto\n", t); return 1; }
      continue;
    }
    if(!strcmp(argv[i], "--wavelengths") && i+1 < argc)
    {
      wavelengths = atoi(argv[++i]);
      if(wavelengths != 1 && wavelengths != MI_WAVELENGTHS_HERO) { fprintf(stderr, "[corona-mi] --wavelengths %d: expected 1 or 4\n", wavelengths); return 1; }
      continue;
    }
    if(!strcmp(argv[i], "-s") && i+1 < argc) spp = strtoull(argv[++i], 0, 10);
    else if(!strcmp(argv[i], "-w") && i+1 < argc) opt.width = atoi(argv[++i]);
    else if(!strcmp(argv[i], "-h") && i+1 < argc) opt.height = atoi(argv[++i]);
    else if(!strcmp(argv[i], "-x") && i+1 < argc) postfix = argv[++i];
    else if(!strcmp(argv[i], "-c") && i+1 < argc) opt.cam_file = argv[++i];
    else if(!strcmp(argv[i], "-t") && i+1 < argc) ++i;
    else if(!strcmp(argv[i], "--frame") && i+1 < argc) opt.frame = strtoull(argv[++i], 0, 10);
    else if(!strcmp(argv[i], "--batch") && i+1 < argc) batch = strtoull(argv[++i], 0, 10);
    else if(!strcmp(argv[i], "--iso") && i+1 < argc) opt.iso = atof(argv[++i]);
    else if(!strcmp(argv[i], "--max-verts") && i+1 < argc) opt.max_verts = atoi(argv[++i]);
    else if(!strcmp(argv[i], "--rgb2spec") && i+1 < argc) opt.rgb2spec_lut = argv[++i];
    else if(!strcmp(argv[i], "--info")) info_only = 1;
    else if(!strcmp(argv[i], "--device-build")) device_build = 1;     /* hand the scene over without the host-built tree */
    else if(!strcmp(argv[i], "--sampler") && i+1 < argc) opt.sampler = !strcmp(argv[++i], "ptdl") ? MI_SAMPLER_PTDL : MI_SAMPLER_PT;
    else if(!strcmp(argv[i], "--pointsampler") && i+1 < argc) opt.pointsampler = !strcmp(argv[++i], "halton") ? MI_POINTS_HALTON : MI_POINTS_RAND;
  }
  if(!batch) batch = 1;
  ch_scene *scene = 0;
  if(ch_scene_load(argv[1], &opt, &scene)) return 2;
  const mi_scene_desc *d = ch_scene_desc(scene);
  if(info_only)
  {
    uint64_t kinds[5] = {0, 0, 0, 0, 0}, leaves = 0, leaf_prims = 0;
    uint32_t leaf_max = 0;
    for(uint64_t i=0;i<d->num_prims;i++) kinds[MI_PRIMID_VCNT(d->primid[i]) <= 4 ? MI_PRIMID_VCNT(d->primid[i]) : 0]++;
    for(uint32_t n=0;n<d->num_nodes;n++) for(int c=0;c<4;c++) if(d->nodes[n].child[c] & MI_NODE_LEAF)
    {
      const uint32_t cnt = (uint32_t)(d->nodes[n].child[c] & 31);
      if(cnt) { leaves++; leaf_prims += cnt; if(cnt > leaf_max) leaf_max = cnt; }
    }
    printf("scene    : %s\n", argv[1]);
    printf("film     : %ux%u (padded to multiples of 32), max path vertices %u, sampler %s, points %s, frame %lu\n", d->width, d->height, d->max_verts,
        d->sampler == MI_SAMPLER_PTDL ? "ptdl" : "pt", d->pointsampler == MI_POINTS_HALTON ? "halton" : "rand", (unsigned long)d->frame);
    printf("shapes   : %u, materials %u\n", d->num_shapes, d->num_materials);
    printf("prims    : %lu (spheres %lu, lines %lu, triangles %lu, quads %lu)\n", (unsigned long)d->num_prims, (unsigned long)kinds[1],
        (unsigned long)kinds[2], (unsigned long)kinds[3], (unsigned long)kinds[4]);
    printf("accel    : qbvh, %u nodes, %lu leaves, %.2f prims per leaf (max %u)\n", d->num_nodes, (unsigned long)leaves,
        leaves ? (double)leaf_prims/leaves : 0.0, leaf_max);
    printf("aabb     : (%.3f, %.3f)x(%.3f, %.3f)x(%.3f, %.3f) dm^3\n", d->aabb[0], d->aabb[3], d->aabb[1], d->aabb[4], d->aabb[2], d->aabb[5]);
    printf("emitters : %u primitives, p_geo %.3f\n", d->lights.num_prims, d->lights.p_geo);
    printf("camera   : pos (%.3f %.3f %.3f) focus %.3f focal length %.4f f/%.2f film %.4fx%.4f\n", d->cam.pos[0], d->cam.pos[1], d->cam.pos[2],
        d->cam.focus, d->cam.focal_length, d->cam.f_stop, d->cam.film_width, d->cam.film_height);
    ch_scene_free(scene);
    return 0;
  }
  mi_group *group = 0;
  mi_scene_desc without_tree = *d;
  if(device_build) { without_tree.nodes = 0; without_tree.num_nodes = 0; d = &without_tree; }
  if(!num_devices)
  { /* one GPU: the one mi_init picks (LOCAL_RANK under a one-process-per-GPU launcher, else device 0) */
    if(mi_init(-1)) { fprintf(stderr, "[main] %s\n", mi_last_error()); return 2; }
    devices[0] = mi_current_device();
    num_devices = 1;
  }
  if(mi_group_create(d, devices, num_devices, &group)) { fprintf(stderr, "[main] %s\n", mi_last_error()); return 2; }
  if(traversal >= 0) for(int k=0;k<num_devices;k++) mi_scene_set_traversal(mi_group_scene(group, k), traversal);
  if(wavelengths != 1) for(int k=0;k<num_devices;k++)
    if(mi_scene_set_wavelengths(mi_group_scene(group, k), wavelengths)) { fprintf(stderr, "[main] %s\n", mi_last_error()); return 2; }
  if(num_devices > 1) printf("[main] %d GPUs, framebuffer reduce: %s\n", num_devices, mi_group_uses_rccl(group) ? "RCCL (ncclReduce)" : "peer copies + add kernel");

  const uint64_t per = (uint64_t)d->width*d->height;
  uint64_t counter = 0, overlays = 0;
  double t_prog = 0.0;
  while(overlays < spp)
  { /* view_render: step the sample counter in batch_frames * W*H intervals (src/view.c:636-638) */
    const uint64_t b = overlays + batch > spp ? spp - overlays : batch;
    const double t0 = now();
    if(mi_group_render(group, counter, b*per) || mi_group_sync(group)) { fprintf(stderr, "[main] %s\n", mi_last_error()); return 3; }
    t_prog += now() - t0;
    counter += b*per; overlays += b;
    printf("  %.3f s/frame, %lu spp      \r", (now() - t0)/b, (unsigned long)overlays);
    fflush(stdout);
  }
  float *fb = (float *)calloc(3*per, sizeof(float));
  const double t0 = now();
  if(mi_group_fb_read(group, fb, 0)) return 3;            /* the framebuffer reduce over the GPUs and the read-back are part of the progression's time */
  t_prog += now() - t0;

  char base[1024], fn[1200];
  snprintf(base, sizeof(base), "%s", argv[1]);
  char *dot = strrchr(base, '.');
  if(dot && !strchr(dot, '/')) *dot = 0;
  snprintf(fn, sizeof(fn), "%s%s_fb00.pfm", base, postfix);
  const float gain = ch_scene_gain(scene, overlays);
  ch_pfm_write(fn, fb, d->width, d->height, gain);
  double mean[3] = {0, 0, 0};
  for(uint64_t i=0;i<per;i++) for(int k=0;k<3;k++) mean[k] += fb[3*i+k];
  uint64_t cnt[8];
  mi_group_counters(group, cnt);
  strncat(fn, ".txt", sizeof(fn) - strlen(fn) - 1);
  FILE *f = fopen(fn, "wb");
  if(f)
  {
    fprintf(f, "corona-mi: MI355X backend for the corona-13 pt/ptdl hot path\nfile     : %s\n", base);
    fprintf(f, "aabb     : (%.3f, %.3f)x(%.3f, %.3f)x(%.3f, %.3f) dm^3\n", d->aabb[0], d->aabb[3], d->aabb[1], d->aabb[4], d->aabb[2], d->aabb[5]);
    fprintf(f, "points   : xorshift128+ per path index\nprimitive: %lu indexed primitives\naccel    : qbvh, %u nodes\n", (unsigned long)d->num_prims, d->num_nodes);
    fprintf(f, "view     : samples per pixel: %lu (%.4f s/prog) max path vertices %u\n           res %ux%u\n           elapsed wallclock prog %.3fs\n",
        (unsigned long)overlays, t_prog/overlays, d->max_verts, d->width, d->height, t_prog);
    fprintf(f, "           cam 0 average image intensity (rgb): (%f %f %f)\n", mean[0]*gain/per, mean[1]*gain/per, mean[2]*gain/per);
    fprintf(f, "sampler  : %s\n", d->sampler == MI_SAMPLER_PTDL ? "pathtracer with next event estimation and mis" : "pathtracer");
    fprintf(f, "mutations: %s\n", d->pointsampler == MI_POINTS_HALTON ? "halton points" : "none");
    if(cnt[0])        /* the debug counters are only live under CORONA_MI_COUNTERS=1 (mi_scene_set_counters), like the reference's -DACCEL_DEBUG */
      fprintf(f, "work     : %.4f rays %.4f node visits %.4f prim tests %.5f splats per sample\n",
          (double)cnt[0]/cnt[4], (double)cnt[1]/cnt[4], (double)cnt[3]/cnt[4], (double)cnt[5]/cnt[4]);
    fclose(f);
  }
  printf("\n[main] rendered %lu spp in %.3f s (%.2f Msamples/s), saved %s%s_fb00.pfm\n", (unsigned long)overlays, t_prog,
      overlays*per/t_prog*1e-6, base, postfix);
  free(fb);
  mi_group_destroy(group);
  mi_shutdown();
  ch_scene_free(scene);
  return 0;
}
