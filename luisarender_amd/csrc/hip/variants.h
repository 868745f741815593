// variants.h — the precompiled megakernel feature masks: every entry of lrd::kSceneVariants (megapath_kernel.h)
// x {counters} x {generic sampler}.  One object file each (megapath_variant.hip, -DLR_VARIANT=<mask>), built in
// parallel by the Makefile (VARIANT_MASKS must list the same numbers).
#pragma once
#define LR_VARIANT_LIST(X)                                                                  \
    X(0) X(1) X(2) X(3)         /* lean: Matte / Mirror / Glass / Plastic / Metal, lights */  \
    X(4) X(5) X(6) X(7)         /* + image / directional / combined environment */           \
    X(8) X(9) X(10) X(11)       /* lean + alpha-tested traversal (round 2: such a scene ran 25 % slower on <60>) */ \
    X(12) X(13) X(14) X(15)     /* + environment + alpha test */                             \
    X(16) X(17) X(18) X(19)     /* + Disney */                                               \
    X(20) X(21) X(22) X(23)     /* + environment + Disney */                                 \
    X(60) X(61) X(62) X(63)     /* + environment + alpha test + Disney + Mix */              \
    X(124) X(125) X(126) X(127) /* everything (+ Layered) */                                 \
    X(636) X(637) X(638) X(639) /* everything + Mix / Layered nested in each other (kFeatNest) */ \
    X(252) X(253) X(254) X(255) /* everything + the sibling integrators Direct / Normal */ \
    X(256) X(257) X(258) X(259) /* the volumetric megakernel MegaVPTNaive (megavpt_kernel.h) */ \
    X(1024) X(1025) X(1026) X(1027) /* wavefront mode, camera pass: the lean kernel that parks heavy hits (kFeatWf) */ \
    X(1028) X(1029) X(1030) X(1031) /* + environment */ \
    X(1032) X(1033) X(1034) X(1035) /* + alpha-tested traversal */ \
    X(1036) X(1037) X(1038) X(1039) /* + environment + alpha test */ \
    X(3072) X(3073) X(3074) X(3075) /* wavefront mode, continuation pass (kFeatWf | kFeatCont) */ \
    X(3076) X(3077) X(3078) X(3079) /* + environment */ \
    X(3080) X(3081) X(3082) X(3083) /* + alpha test */ \
    X(3084) X(3085) X(3086) X(3087) /* + environment + alpha test */ \
    /* round 4: the path-pool scheduler (megapool_kernel.h, kFeatPool = 4096) for the lean masks ... */ \
    X(4096) X(4097) X(4098) X(4099) X(4100) X(4101) X(4102) X(4103) X(4104) X(4105) X(4106) X(4107) \
    X(4108) X(4109) X(4110) X(4111) X(4112) X(4113) X(4114) X(4115) X(4116) X(4117) X(4118) X(4119) \
    /* ... the wavefront camera pass ... */ \
    X(5120) X(5121) X(5122) X(5123) X(5124) X(5125) X(5126) X(5127) X(5128) X(5129) X(5130) X(5131) X(5132) X(5133) X(5134) X(5135) \
    /* ... and the continuation pass */ \
    X(7168) X(7169) X(7170) X(7171) X(7172) X(7173) X(7174) X(7175) X(7176) X(7177) X(7178) X(7179) X(7180) X(7181) X(7182) X(7183) \
    /* round 6: lean kernels that decode 8-bit texels (kFeatByteTex = 8192): Disney, environment + Disney, one path per lane and pool */ \
    X(8208) X(8209) X(8210) X(8211) X(8212) X(8213) X(8214) X(8215) X(12304) X(12305) X(12306) X(12307) X(12308) X(12309) X(12310) X(12311)

// round 6: the lean pool kernels once more with the generic sampler's kind fixed to PaddedSobol (kFeatPadded = 16384 | kFeatGeneric): plain,
// environment, alpha, environment + alpha, Disney, environment + Disney, and the 8-bit-texel Disney sets, each with its counting twin.  Not part of
// the kSceneVariants x {Count} x {Generic} grid: lrhip_render looks them up by mask (kPaddedVariants).  Measured (profiles/r06y_padded_sobol_kernels.txt,
// r06za_padded_draws_out_of_line.txt, r06zf_padded_environment_set.txt): C2 949 -> 1004 Msamples/s at 256 spp, the camera class under PaddedSobol 1007 -> 1111,
// the bedroom class 974 -> 1011.
#define LR_PADDED_LIST(X) X(20482) X(20483) X(20486) X(20487) X(20490) X(20491) X(20494) X(20495) X(20498) X(20499) X(20502) X(20503) X(28690) X(28691) X(28694) X(28695) \
    /* ... and the lean passes of wavefront mode, camera pass and continuation pass, plain and with the alpha-tested traversal (the kitchen class under PaddedSobol) */ \
    X(21506) X(21507) X(21514) X(21515) X(23554) X(23555) X(23562) X(23563)
