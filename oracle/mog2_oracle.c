/* TEST INFRASTRUCTURE ONLY (never linked or loaded by the product package).
 *
 * CPU restatement of the background model of the reference's IR tracker, SURVEY section 8 f4:
 *   CVBackground (track/cliptracker.py:561-613) = cv2.createBackgroundSubtractorMOG2(history=1000, detectShadows=False),
 *   .apply(frame, None, learning_rate) per frame, .getBackgroundImage().
 * The algorithm lives in a third-party dependency that is not vendored: opencv-contrib-python-headless~=4.12.0.88
 * (requirements.txt:4), modules/video/src/bgfg_gaussmix2.cpp (Zivkovic, "Improved adaptive Gaussian mixture model for
 * background subtraction", ICPR 2004; Zivkovic & van der Heijden, PRL 2006).  This file restates the published
 * per-pixel update for 8-bit single-channel frames with that implementation's defaults (5 mixtures, background
 * ratio 0.9, generation threshold 9, initial / min / max variance 15 / 4 / 75, complexity reduction 0.05) and its
 * learning-rate rule (1 / min(2 * nframes, history) when the caller passes a negative rate or on the first frame).
 *
 * PARITY UNPINNED: cv2 is not installed in the build container and the reference holds no golden vector for this stage,
 * so nothing here has been compared with OpenCV's output.  Float arithmetic is single precision, one operation at a
 * time (build with -ffp-contract=off), in the order written. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct mog2 {
  int w, h, nmix, nframes, history;
  float var_threshold, background_ratio, var_threshold_gen, var_init, var_min, var_max, ct;
  float *weight, *var, *mean; /* [pixel][mode] */
  unsigned char* modes;       /* modes in use per pixel */
} mog2;

mog2* mog2_create(int w, int h, int history, float var_threshold) {
  mog2* m = (mog2*)calloc(1, sizeof(mog2));
  m->w = w; m->h = h; m->nmix = 5; m->history = history > 0 ? history : 500;
  m->var_threshold = var_threshold; m->background_ratio = 0.9f; m->var_threshold_gen = 9.0f;
  m->var_init = 15.0f; m->var_min = 4.0f; m->var_max = 75.0f; m->ct = 0.05f;
  const size_t n = (size_t)w * h * m->nmix;
  m->weight = (float*)calloc(n, sizeof(float));
  m->var = (float*)calloc(n, sizeof(float));
  m->mean = (float*)calloc(n, sizeof(float));
  m->modes = (unsigned char*)calloc((size_t)w * h, 1);
  return m;
}
void mog2_destroy(mog2* m) {
  if (!m) return;
  free(m->weight); free(m->var); free(m->mean); free(m->modes); free(m);
}
int mog2_nframes(const mog2* m) { return m->nframes; }

/* the learning rate the update of the next frame will use, as apply() derives it */
double mog2_rate(const mog2* m, double learning_rate) {
  const int nframes = m->nframes + 1;
  const int lim = 2 * nframes < m->history ? 2 * nframes : m->history;
  return (learning_rate >= 0 && nframes > 1) ? learning_rate : 1.0 / lim;
}

void mog2_apply(mog2* m, const unsigned char* img, double learning_rate, unsigned char* mask) {
  const double rate = mog2_rate(m, learning_rate);
  m->nframes += 1;
  const float alphaT = (float)rate;
  const float alpha1 = 1.0f - alphaT;
  const float prune = (float)(-rate * m->ct);
  const float Tb = m->var_threshold, TB = m->background_ratio, Tg = m->var_threshold_gen;
  const int K = m->nmix;
  for (size_t p = 0; p < (size_t)m->w * m->h; ++p) {
    float* wgt = m->weight + p * K;
    float* var = m->var + p * K;
    float* mean = m->mean + p * K;
    const float data = (float)img[p];
    int background = 0, fits = 0;
    int nmodes = m->modes[p];
    float total = 0.0f;
    for (int mode = 0; mode < nmodes; ++mode) {
      float weight = alpha1 * wgt[mode] + prune;
      int swap_count = 0;
      if (!fits) {
        const float v = var[mode];
        const float d = mean[mode] - data;
        const float dist2 = d * d;
        if (total < TB && dist2 < Tb * v) background = 1;
        if (dist2 < Tg * v) {
          fits = 1;
          weight += alphaT;
          const float k = alphaT / weight;
          mean[mode] -= k * d;
          float varnew = v + k * (dist2 - v);
          varnew = varnew > m->var_min ? varnew : m->var_min;
          varnew = varnew < m->var_max ? varnew : m->var_max;
          var[mode] = varnew;
          for (int i = mode; i > 0; --i) { /* the matched mode moves up past lighter ones */
            if (weight < wgt[i - 1]) break;
            ++swap_count;
            float t;
            t = wgt[i]; wgt[i] = wgt[i - 1]; wgt[i - 1] = t;
            t = var[i]; var[i] = var[i - 1]; var[i - 1] = t;
            t = mean[i]; mean[i] = mean[i - 1]; mean[i - 1] = t;
          }
        }
      }
      if (weight < -prune) {
        weight = 0.0f;
        --nmodes;
      }
      wgt[mode - swap_count] = weight;
      total += weight;
    }
    float inv = 0.0f;
    if (fabsf(total) > FLT_EPSILON) inv = 1.0f / total;
    for (int mode = 0; mode < nmodes; ++mode) wgt[mode] *= inv;
    if (!fits && alphaT > 0.0f) {
      const int mode = nmodes == K ? K - 1 : nmodes++;
      if (nmodes == 1) {
        wgt[mode] = 1.0f;
      } else {
        wgt[mode] = alphaT;
        for (int i = 0; i < nmodes - 1; ++i) wgt[i] *= alpha1;
      }
      mean[mode] = data;
      var[mode] = m->var_init;
      for (int i = nmodes - 1; i > 0; --i) {
        if (alphaT < wgt[i - 1]) break;
        float t;
        t = wgt[i]; wgt[i] = wgt[i - 1]; wgt[i - 1] = t;
        t = var[i]; var[i] = var[i - 1]; var[i - 1] = t;
        t = mean[i]; mean[i] = mean[i - 1]; mean[i - 1] = t;
      }
    }
    m->modes[p] = (unsigned char)nmodes;
    mask[p] = background ? 0 : 255;
  }
}

/* getBackgroundImage: weighted mean of the heaviest modes up to the background ratio, rounded to nearest even */
void mog2_background(const mog2* m, unsigned char* out) {
  const int K = m->nmix;
  for (size_t p = 0; p < (size_t)m->w * m->h; ++p) {
    const int nmodes = m->modes[p];
    float mean_val = 0.0f, total = 0.0f;
    for (int mode = 0; mode < nmodes; ++mode) {
      const float w = m->weight[p * K + mode];
      mean_val += w * m->mean[p * K + mode];
      total += w;
      if (total > m->background_ratio) break;
    }
    float inv = 0.0f;
    if (fabsf(total) > FLT_EPSILON) inv = 1.0f / total;
    mean_val *= inv;
    long r = lrintf(mean_val);
    out[p] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
  }
}

/* state export for comparisons: [pixel][mode] arrays as they are */
void mog2_state(const mog2* m, float* weight, float* var, float* mean, unsigned char* modes) {
  const size_t n = (size_t)m->w * m->h * m->nmix;
  memcpy(weight, m->weight, n * sizeof(float));
  memcpy(var, m->var, n * sizeof(float));
  memcpy(mean, m->mean, n * sizeof(float));
  memcpy(modes, m->modes, (size_t)m->w * m->h);
}
