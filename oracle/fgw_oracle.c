/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See fgw_oracle_impl.h for the contract.
 * Builds two instantiations of the CPU restatement: *_f32 (REAL=float) and *_f64 (REAL=double).
 * Built by oracle/Makefile into oracle/libconan_oracle.so (git-ignored; travels to the GPU box).
 */
#include <math.h>
#include <stdlib.h>

static float r_exp_f32(float x) { return expf(x); }
static float r_log_f32(float x) { return logf(x); }
static float r_sqrt_f32(float x) { return sqrtf(x); }
static double r_exp_f64(double x) { return exp(x); }
static double r_log_f64(double x) { return log(x); }
static double r_sqrt_f64(double x) { return sqrt(x); }

#define REAL float
#define SUF _f32
#include "fgw_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF _f64
#include "fgw_oracle_impl.h"
#undef REAL
#undef SUF
