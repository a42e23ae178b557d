/*
 * oracle_cli.c -- command-line driver of the CPU restatement (TEST INFRASTRUCTURE ONLY).
 *
 *   qcd_oracle_cli <init.dat> [--rng seq|persite] [--time-sweeps N]
 *
 * Reads the reference's init.dat (tokenizer restated from SU(N)/suncpp/IO/io.cpp:42-116,
 * key table from SU(N)/suncl/suncl.cpp:305-459, defaults from SU(N)/QCDGPU.cpp:98-220),
 * runs the sequential simulation (SU(N)/suncpp/suncpp.h:73-165) and prints the data table
 * in the reference's "% 16.13e" format (SU(N)/suncl/suncl.cpp:1822-1850).
 * With --time-sweeps N it instead times N heat-bath sweeps after initialisation and prints
 * one JSON line with link-updates/s on one core (the cpu_baseline leg of bench.py).
 */
#include "qcd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define LENGTH 250

typedef struct {
    int group, nd, L[4], iter, niter, nhit, nav, ints, lux;
    double beta;
    unsigned randseries;
} cfg_t;

static void defaults(cfg_t *c)
{
    /* SU(N)/QCDGPU.cpp:98-189 */
    c->group = 3; c->nd = 4;
    c->L[0] = c->L[1] = c->L[2] = c->L[3] = 32;
    c->iter = 2; c->niter = 1; c->nhit = 10; c->nav = 0; c->ints = 1;
    c->beta = 6.5; c->lux = 3; c->randseries = 54;
}

static int parse(const char *fname, cfg_t *c)
{
    FILE *f = fopen(fname, "r");
    if (!f) return -1;
    char line[LENGTH], var[LENGTH], txt[LENGTH];
    while (fgets(line, LENGTH, f) != NULL) {
        int s1 = 0;
        while (line[s1] == ' ' || line[s1] == '\t') s1++;
        size_t eq = strcspn(line, "=");
        if (eq >= strlen(line)) continue;
        int e = (int)eq - 1;
        while (e >= 0 && (line[e] == ' ' || line[e] == '\t')) e--;
        int n = 0;
        for (int j = s1; j <= e; j++) var[n++] = line[j];
        var[n] = 0;
        int fin = (int)strcspn(line, "#") - 1;
        while (fin >= 0 && (line[fin] == ' ' || line[fin] == '\t' || line[fin] == '\n' || line[fin] == '\r')) fin--;
        int s3 = (int)eq + 1;
        while (line[s3] == ' ' || line[s3] == '\t') s3++;
        n = 0;
        for (int j = s3; j <= fin; j++) txt[n++] = line[j];
        txt[n] = 0;
        int iv = 0; double fv = 0.0;
        sscanf(txt, "%d", &iv);
        sscanf(txt, "%lf", &fv);
        if (!strcmp(var, "GROUP")) c->group = iv;
        if (!strcmp(var, "ND")) c->nd = iv;
        if (!strcmp(var, "L1")) c->L[0] = iv;
        if (!strcmp(var, "L2")) c->L[1] = iv;
        if (!strcmp(var, "L3")) c->L[2] = iv;
        if (!strcmp(var, "L4")) c->L[3] = iv;
        if (!strcmp(var, "LS")) c->L[0] = c->L[1] = c->L[2] = iv;
        if (!strcmp(var, "LT")) c->L[c->nd - 1] = iv;
        if (!strcmp(var, "ITER")) c->iter = iv;
        if (!strcmp(var, "NITER")) c->niter = iv;
        if (!strcmp(var, "NHIT")) c->nhit = iv;
        if (!strcmp(var, "BETA")) c->beta = fv;
        if (!strcmp(var, "NAV")) c->nav = iv;
        if (!strcmp(var, "INTS")) c->ints = iv;
        if (!strcmp(var, "RANDSERIES")) c->randseries = (unsigned)iv;
        if (!strcmp(var, "PRNG")) {
            if (!strcmp(txt, "RANLUX0")) c->lux = 0;
            if (!strcmp(txt, "RANLUX1")) c->lux = 1;
            if (!strcmp(txt, "RANLUX2")) c->lux = 2;
            if (!strcmp(txt, "RANLUX3")) c->lux = 3;
            if (!strcmp(txt, "RANLUX4")) c->lux = 4;
        }
    }
    fclose(f);
    return 0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
    cfg_t c;
    defaults(&c);
    int rng = QO_RNG_SEQ, time_sweeps = 0;
    const char *file = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--rng") && i + 1 < argc) {
            i++;
            if (!strcmp(argv[i], "persite")) rng = QO_RNG_PERSITE;
        } else if (!strcmp(argv[i], "--time-sweeps") && i + 1 < argc) {
            time_sweeps = atoi(argv[++i]);
        } else {
            file = argv[i];
        }
    }
    if (file && parse(file, &c) != 0) {
        fprintf(stderr, "cannot read %s\n", file);
        return 2;
    }
    if (c.ints == 2) rng = QO_RNG_GID;
    qo_lattice *lat = qo_create(c.group, c.L, c.beta, c.nhit, c.ints, rng, c.lux, c.randseries);

    if (time_sweeps > 0) {
        qo_init_links(lat);
        qo_sweep(lat);                       /* warm-up, untimed */
        double t0 = now_s();
        for (int s = 0; s < time_sweeps; s++) qo_sweep(lat);
        double dt = now_s() - t0;
        double plq[3];
        qo_plaquette(lat, plq);
        printf("{\"link_updates_per_s\": %.6e, \"sweeps\": %d, \"seconds\": %.4f, \"sites\": %d, "
               "\"group\": %d, \"plq_total\": %.13e}\n",
               4.0 * lat->sites * time_sweeps / dt, time_sweeps, dt, lat->sites, c.group, plq[2]);
        qo_destroy(lat);
        return 0;
    }

    double *plq = (double *)calloc(3 * (size_t)c.iter, sizeof(double));
    double *act = (double *)calloc(3 * (size_t)c.iter, sizeof(double));
    qo_simulate(lat, c.nav, c.iter, c.niter, plq, act);
    printf("    #,  %-21s %-21s %-21s %-21s %-21s %-21s \n", "Plq_spat", "Plq_temp", "Plq_total",
           "S_spat", "S_temp", "S_total");
    for (int i = 0; i < c.iter; i++)
        printf("%5i % 16.13e % 16.13e % 16.13e % 16.13e % 16.13e % 16.13e\n", i, plq[3 * i],
               plq[3 * i + 1], plq[3 * i + 2], act[3 * i], act[3 * i + 1], act[3 * i + 2]);
    free(plq);
    free(act);
    qo_destroy(lat);
    return 0;
}
