// quantify_mef.cpp -- C++ consumer of include/canopy-cpp.h (the reference's examples/ target is an
// empty CMakeLists.txt): Open-PSA MEF fault tree -> minimal cut sets -> rare-event / MCUB on the GPU,
// plus a Monte-Carlo run through the gate DAG.
//   hipcc -std=c++17 -Iinclude examples/quantify_mef.cpp -Lcanopy_amd -lcanopy -Wl,-rpath,$PWD/canopy_amd -o examples/quantify_mef
//   ./examples/quantify_mef examples/two_of_three_plus_or.xml
#include <cstdio>
#include <vector>

#include "canopy-cpp.h"

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "examples/two_of_three_plus_or.xml";
    char err[512] = {0};
    canopy_ft* ft_raw = nullptr;
    if (canopy_ft_parse_file(path, &ft_raw, err, sizeof(err)) != CANOPY_OK) {
        std::fprintf(stderr, "cannot read %s: %s\n", path, err);
        return 2;
    }
    std::unique_ptr<canopy_ft, void (*)(canopy_ft*)> ft(ft_raw, canopy_ft_free);
    canopy_ft_info info;
    canopy_ft_get_info(ft.get(), &info);
    std::vector<double> p((size_t)info.n_basic_events);
    canopy_ft_probabilities(ft.get(), p.data());
    int64_t n_sets = 0, nnz = 0;
    if (canopy_ft_cut_sets(ft.get(), 20, &n_sets, &nnz, nullptr, nullptr) != CANOPY_OK) {
        std::fprintf(stderr, "the tree is not coherent: no cut sets\n");
        return 3;
    }
    std::vector<int64_t> row_ptr((size_t)n_sets + 1);
    std::vector<int32_t> col_idx((size_t)nnz);
    canopy_ft_cut_sets(ft.get(), 20, &n_sets, &nnz, row_ptr.data(), col_idx.data());
    try {
        canopy::context ctx(0);
        auto model = ctx.upload(n_sets, info.n_basic_events, row_ptr.data(), col_idx.data());
        ctx.set_probabilities(model, p);
        canopy::quant_params prm;
        prm.cut_off = 0.0;
        const canopy_quant_result r = ctx.quantify(model, prm);
        // uncertainty trials: the same cut sets under three probability vectors in one call (trial 1 = point values)
        std::vector<double> trials;
        for (double f : {0.5, 1.0, 2.0})
            for (double pj : p) trials.push_back(pj * f > 1.0 ? 1.0 : pj * f);
        const std::vector<canopy_quant_result> rb = ctx.quantify_batch(model, trials, 3, prm);
        // Monte-Carlo through the same tree's gate DAG
        std::vector<int32_t> gt((size_t)info.n_gates), gm((size_t)info.n_gates), ap((size_t)info.n_gates + 1),
            ar((size_t)(info.n_dag_args > 0 ? info.n_dag_args : 1));
        int32_t top = 0;
        canopy_ft_dag(ft.get(), gt.data(), gm.data(), ap.data(), ar.data(), &top);
        canopy_dag_view dv{info.n_basic_events, info.n_gates, gt.data(), gm.data(), ap.data(), ar.data(), top, 0};
        canopy_dag* dag_raw = nullptr;
        if (canopy_dag_upload(ctx.get(), &dv, &dag_raw) != CANOPY_OK) throw canopy::error(CANOPY_ERR_INVALID_MODEL, "canopy_dag_upload", canopy_last_error(ctx.get()));
        std::unique_ptr<canopy_dag, void (*)(canopy_dag*)> dag(dag_raw, canopy_dag_free);
        canopy_dag_set_probabilities(ctx.get(), dag.get(), p.data());
        canopy_mc_params mp{0 /*seed: settings.h:324*/, 0, 10000000};
        canopy_mc_result mc;
        if (canopy_mc_run(ctx.get(), dag.get(), &mp, &mc) != CANOPY_OK) throw canopy::error(CANOPY_ERR_HIP, "canopy_mc_run", canopy_last_error(ctx.get()));
        std::printf("{\"model\": \"%s\", \"top\": \"%s\", \"basic_events\": %d, \"cut_sets\": %lld, \"rare_event\": %.17g, "
                    "\"mcub\": %.17g, \"batch_rare_event\": [%.17g, %.17g, %.17g], \"monte_carlo\": %.9g, \"mc_std_error\": %.3g, "
                    "\"mc_trials\": %lld}\n",
                    path, canopy_ft_gate_name(ft.get(), info.top_gate), info.n_basic_events, (long long)n_sets,
                    r.rare_event, r.mcub, rb[0].rare_event, rb[1].rare_event, rb[2].rare_event, mc.estimate, mc.std_error,
                    (long long)mc.num_trials);
    } catch (const canopy::error& e) {
        std::fprintf(stderr, "canopy: %s\n", e.what());
        return 1;
    }
    return 0;
}
