// Dev tool (host): statistics of the PRODUCT's AdjStaged::soil (csrc/hbv_adj_step.h) on bench-shaped forcing: updates per
// lane-day and per wave-day (64 lanes = 4 basins x 16 members), true |G2| (float64) at the returned state.
//   g++ -O2 -std=c++17 -ffp-contract=off -Ihydrodl2_amd/csrc -o /tmp/adj_soil_stats tools/micro/adj_soil_stats.cpp
//   (-DADJ_SOIL_HH=0 -DADJ_SOIL_LIN=0.0f -DADJ_SOIL_KINK=0: the round-3 solve)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "hbv_adj_step.h"
using namespace hbvx;
int main()
{
    const int T = 7300, B = 64, M = 16;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(0, 1);
    std::normal_distribution<float> Nn(0, 1);
    const float lo[13] = {1, 50, .05, .01, .001, .2, 0, 0, -2.5, .5, 0, 0, .3}, hi[13] = {6, 1000, .9, .5, .2, 1, 10, 100, 2.5, 10, .1, .2, 5};
    std::vector<float> P(T * B), Tm(T * B), PET(T * B), boff(B);
    for (int b = 0; b < B; b++) boff[b] = U(rng) * 25 - 10;
    for (int t = 0; t < T; t++) for (int b = 0; b < B; b++) {
        float season = sinf(2 * M_PI * t / 365.0f);
        P[t * B + b] = fmaxf((U(rng) - 0.7f) * 60.0f, 0.0f);
        Tm[t * B + b] = 10 * season + 5 * Nn(rng) + boff[b];
        PET[t * B + b] = fmaxf(3 + 2.5f * season + 0.3f * Nn(rng), 0.0f);
    }
    std::vector<float> par(B * M * NPARAM_MAX, 0.f), st(B * M * 5, 0.f);
    for (int n = 0; n < B * M; n++) for (int i = 0; i < 13; i++) {
        float u = 1.0f / (1.0f + expf(-Nn(rng)));
        par[n * NPARAM_MAX + i] = u * (hi[i] - lo[i]) + lo[i];
    }
    long nall = 0, waves = 0, hist[8] = {0}, whist[8] = {0}, bad = 0;
    double gmax = 0;
    for (int t = 0; t < T; t++)
        for (int w = 0; w < B / 4; w++) {
            int wmax = 0;
            for (int l = 0; l < 64; l++) {
                int b = w * 4 + l / 16, n = b * M + (l % 16);
                float *p = &par[n * NPARAM_MAX], *s = &st[n * 5];
                float u = 1.0f / (1.0f + expf(-Nn(rng)));
                p[P_BETAET] = u * (5 - .3f) + .3f;
                float y0, y1, rf, Is, y2, Peff, ex, y3, y4, Q;
                AdjStaged<true>::snow(p, P[t * B + b], Tm[t * B + b], 1.0f, s[0], s[1], y0, y1, rf, Is);
                int it = AdjStaged<true>::soil(p, rf, Is, PET[t * B + b], 1.0f, s[2], 1e-3f, 3, y2, Peff, ex);
                AdjStaged<true>::gw(p, Peff, ex, 1.0f, s[3], s[4], y3, y4, Q);
                // true residual at the returned state (double)
                double SM = fmax(y2, 1e-8), sw = fmin(pow(SM / p[P_FC], p[P_BETA]), 1.0), ef = fmin(pow(SM / (p[P_LP] * p[P_FC]), p[P_BETAET]), 1.0);
                double G2 = (y2 - s[2]) - ((rf + Is) - (rf + Is) * sw - fmax(SM - p[P_FC], 0.0) - fmin(SM, PET[t * B + b] * ef));
                if (fabs(G2) > gmax) gmax = fabs(G2);
                if (fabs(G2) > 1.02e-3) bad++;
                nall++; hist[it < 7 ? it : 7]++;
                wmax = it > wmax ? it : wmax;
                s[0] = y0; s[1] = y1; s[2] = y2; s[3] = y3; s[4] = y4;
            }
            waves++; whist[wmax < 7 ? wmax : 7]++;
        }
    printf("lane-days %ld, wave-days %ld; true |G2| max %.3e, above gtol %ld\n", nall, waves, gmax, bad);
    printf("updates per lane-day:"); for (int i = 0; i < 6; i++) printf(" %d: %.4f%%", i, 100.0 * hist[i] / nall); printf("\n");
    printf("updates per wave-day:"); for (int i = 0; i < 6; i++) printf(" %d: %.3f%%", i, 100.0 * whist[i] / waves); printf("\n");
}
