/* battle_main.c -- TEST INFRASTRUCTURE: runs the reference's own BSDF battle test (tools/battle-test.c:57-266, the thing
 * regression/0052_dielectric and 0053_dielectric are) from the source where it lies under /root/reference.
 *
 * The tool as shipped dereferences rt.pointsampler without ever setting it (tools/battle-test.c:272 vs
 * src/pointsampler.d/rand.c:50-53; the reference's tools/Makefile:18 carries a FIXME) and segfaults. This wrapper is the
 * two-line fix SURVEY section 4 describes, applied from outside: the reference file is compiled unmodified (its main() renamed by
 * the preprocessor), our main() sets the two globals first. Nothing of the reference is copied into the repository.
 *
 *   echo "1.7 73 #" | oracle/_ref/battle_test oracle/_ref/shaders_mv32/libdielectric.so 0.4 <reflect> 4
 *   -> four lines "ebsdf-bsdf-epdf-pdf[k] <sample estimate of the bsdf integral> <eval integral> <histogram pdf> <pdf integral>"
 * (tests/golden/make_battle_golden.py turns them into tests/golden/battle.json). */
#define main reference_battle_test_main
#include "tools/battle-test.c"
#undef main

int main(int argc, char *argv[])
{
  rt.num_threads = 1;
  rt.pointsampler = pointsampler_init(0);
  return reference_battle_test_main(argc, argv);
}
