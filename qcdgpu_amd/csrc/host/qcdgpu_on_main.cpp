// qcdgpu_on_main.cpp -- program entry of the O(N) scalar model with the reference's shape
// (O(N)/QCDGPU.cpp:75-241): hard-coded defaults, argv[1] = init file overriding them, "-KEY=value"
// arguments overriding single keys, then
// lattice_init -> lattice_simulate -> lattice_analysis -> lattice_write_results -> lattice_print_measurements.
//
//   qcdgpu_on [init.dat] [-KEY=value ...] [--profile]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "on_model.h"

int main(int argc, char** argv)
{
    using model_HIP::on_model;
    on_model* model0 = new on_model();
    try {
        std::string overrides;
        for (int i = 1; i < argc; i++) {
            if (!strcmp(argv[i], "--profile")) model0->profiling = true;
            else if (argv[i][0] == '-' && strchr(argv[i], '=')) overrides += std::string(argv[i] + 1) + "\n";   // QCDGPU.cpp:171-192
            else model0->lattice_get_init_file(argv[i]);
        }
        if (!overrides.empty()) {
            // single keys go through the same parser as the init file
            char tmpl[] = "/tmp/qcdgpu_on_XXXXXX";
            const int fd = mkstemp(tmpl);
            if (fd < 0) throw model_HIP::model_error("cannot create a temporary file for the command-line keys");
            FILE* f = fdopen(fd, "w");
            fputs(overrides.c_str(), f);
            fclose(f);
            model0->lattice_get_init_file(tmpl);
            remove(tmpl);
        }
        model0->lattice_init();
        printf("%s\n", model0->header.c_str());
        model0->lattice_simulate();
        model0->lattice_analysis();
        model0->lattice_write_results();
        model0->lattice_print_measurements();
        printf("result file: %s\n", model0->result_file.c_str());
    } catch (const model_HIP::model_error& e) {
        fprintf(stderr, "qcdgpu_on: %s\n", e.what());
        delete model0;
        return 1;
    }
    delete model0;
    printf("\n[...Program is completed successfully...]\n\n");
    return 0;
}
