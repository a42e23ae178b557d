// qcdgpu_main.cpp -- program entry with the reference's shape (SU(N)/QCDGPU.cpp:61-269):
// hard-coded defaults, argv[1] = init.dat overriding them, then
// lattice_init -> lattice_simulate -> lattice_analysis -> lattice_write_results -> lattice_print_measurements.
//
//   qcdgpu [init.dat] [--profile] [--save-state [--save-every SECONDS]] [--stop-at SAMPLE]
//   qcdgpu [init.dat] --load FILE.qcg   continue the run FILE.qcg (in OUTPUTPATH) was saved from
//   qcdgpu --parse-only init.dat     print the run parameters as JSON, touch no device
//   qcdgpu --stats v0 v1 v2 ...      mean and variance of a series (sample 0 excluded)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "model.h"

int main(int argc, char** argv)
{
    using model_HIP::model;
    model* model0 = new model();
    try {
        // defaults (SU(N)/QCDGPU.cpp:98-220)
        model0->lattice_nd = 4;
        model0->lattice_group = 3;
        for (int i = 0; i < 4; i++) model0->lattice_full_size[i] = model0->lattice_domain_size[i] = 32;
        // SU(N)/QCDGPU.cpp:178-179 picks the precision in the source; here it is the library the program is linked with
        model0->precision = (qcd_precision() == 2) ? model::model_precision_double : model::model_precision_single;
        model0->INIT = 1;
        model0->NAV = 0;
        model0->ITER = 2;
        model0->NITER = 1;
        model0->NHIT = 10;
        model0->NHITPar = 1;
        model0->BETA = 6.5;
        model0->ints = model::model_start_cold;
        model0->PL_level = 3;   // SU(N)/QCDGPU.cpp:191: loops, moments and the plane-resolved loops

        const char* init = nullptr;
        bool parse_only = false;
        for (int i = 1; i < argc; i++) {
            if (!strcmp(argv[i], "--profile")) model0->profiling = true;
            else if (!strcmp(argv[i], "--save-state")) model0->turnoff_config_save = false;
            else if (!strcmp(argv[i], "--save-every") && i + 1 < argc) model0->write_lattice_state_every_secs = atof(argv[++i]);
            else if (!strcmp(argv[i], "--stop-at") && i + 1 < argc) model0->stop_at_sample = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--load") && i + 1 < argc) {
                model0->INIT = 0;   // continue (SU(N)/QCDGPU.cpp:180, 234-235)
                model0->fstate = argv[++i];
            }
            else if (!strcmp(argv[i], "--parse-only")) parse_only = true;
            else if (!strcmp(argv[i], "--stats")) {
                model_HIP::data_analysis d;
                for (int k = i + 1; k < argc; k++) d.data.push_back(atof(argv[k]));
                model_HIP::lattice_data_analysis(&d);
                printf("{\"mean\": %.17g, \"variance\": %.17g, \"last\": %.17g}\n", d.mean_value, d.variance, d.GPU_last_value);
                delete model0;
                return 0;
            } else init = argv[i];
        }
        if (init) model0->lattice_get_init_file(const_cast<char*>(init));
        if (parse_only) {
            printf("{\"GROUP\": %d, \"ND\": %d, \"L\": [%d, %d, %d, %d], \"ITER\": %d, \"NITER\": %d, \"NHIT\": %d, "
                   "\"NAV\": %d, \"BETA\": %.17g, \"PHI\": %.17g, \"OMEGA\": %.17g, \"INTS\": %d, \"LUX\": %d, "
                   "\"RANDSERIES\": %u, \"DEVICE\": %d, \"OUTPUTPATH\": \"%s\", \"FINISHPATH\": \"%s\", "
                   "\"PREFIX\": \"%s\", \"PL_LEVEL\": %d, \"WILSONR\": %d, \"WILSONT\": %d, \"GETWILSON\": %d, "
                   "\"NPARTS\": %d, \"N1\": [",
                   model0->lattice_group, model0->lattice_nd, model0->lattice_full_size[0], model0->lattice_full_size[1],
                   model0->lattice_full_size[2], model0->lattice_full_size[3], model0->ITER, model0->NITER, model0->NHIT,
                   model0->NAV, model0->BETA, model0->PHI, model0->OMEGA,
                   model0->ints == model::model_start_hot ? 0 : (model0->ints == model::model_start_cold ? 1 : 2),
                   model0->PRNG0->lux_level(), model0->PRNG0->PRNG_randseries, model0->desired_device,
                   model0->path.c_str(), model0->finishpath.c_str(), model0->fprefix.c_str(), model0->PL_level,
                   model0->wilson_R, model0->wilson_T, (int)model0->get_wilson_loop, model0->lattice_Nparts);
            for (size_t k = 0; k < model0->SubLat.size(); k++) printf("%s%d", k ? ", " : "", model0->SubLat[k].Nx);
            printf("], \"DEVICES\": [");
            for (size_t k = 0; k < model0->SubLat.size(); k++) printf("%s%d", k ? ", " : "", model0->SubLat[k].desired_device);
            printf("]}\n");
            delete model0;
            return 0;
        }

        time_t timer1, timer2;
        time(&timer1);
        model0->lattice_init();
        printf("%s\n", model0->header.c_str());
        model0->lattice_simulate();
        if (!model0->state_file.empty()) printf("state file: %s\n", model0->state_file.c_str());
        if (model0->stopped_early) {
            printf("stopped after sample %d as asked; continue with --load\n", model0->ITER_counter - 1);
            delete model0;
            return 0;
        }
        model0->lattice_analysis();
        model0->lattice_write_results();
        model0->lattice_print_measurements();
        model0->print_time_detailed();
        time(&timer2);
        printf("\nEXECUTION TIME: %f sec\n", difftime(timer2, timer1));
        printf("result file: %s\n", model0->result_file.c_str());
    } catch (const model_HIP::model_error& e) {
        fprintf(stderr, "qcdgpu: %s\n", e.what());
        delete model0;
        return 1;
    }
    delete model0;
    printf("\n[...Program is completed successfully...]\n\n");
    return 0;
}
