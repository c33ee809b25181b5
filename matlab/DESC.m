%% DESC -- drop-in replacement of the reference's Algorithms/DESC.m:14 (called at
%% Demo/compare_algorithms.m:72):   [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
%% PGD (DESC.m:16-261) -> GCW (:263) -> reweighted LAA refinement (:265-313), all on the MI355X; Ind / RijMat go to
%% the GPU once for the three stages (desc_amd_mex('desc', ...): desc_problem_upload).
function [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
    G = params.Gradient;
    make_plots = isfield(params, 'make_plots') && params.make_plots;
    if make_plots || (isa(G, 'HybridGradient') && G.strategy == 0)
        % Adam keeps per-cycle state in the handle object, make_plots needs the per-iteration traces: stage by stage
        [S_vec, traces] = DESC_PGD(Ind, RijMat, params);
        [IndS, perm] = sortrows(double(Ind), [1 2]);
        R_init = desc_amd_mex('gcw', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm));
        disp('Rotation Initialized!')                 % DESC.m:283
        disp('Start DESC refinement ...')             % DESC.m:284
        R_est = desc_amd_mex('refine', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm), R_init);
        disp('DONE!')                                 % DESC.m:313
        if make_plots                                 % the 2 x 2 convergence figure of DESC.m:315-344
            names = {'svec_errors', 'obj_vals', 'MSE_means', 'MSE_medians'};
            labels = {'Average distance to true corruption', 'Value of Objective Function', ...
                      'Mean Error in R estimate (degrees)', 'Median Error in R estimate (degrees)'};
            figure;
            for q = 1:4
                subplot(2, 2, q); plot(1:numel(traces.(names{q})), traces.(names{q}));
                xlabel('Iteration number'); ylabel(labels{q});
            end
        end
        return
    end
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    opt.iters = double(params.iters); opt.decay_interval = 25; opt.hybrid_strategy = 0; opt.t0 = 0;
    switch class(G)
        case 'ConstantStepSize',  opt.step_kind = 0; opt.lr = G.learning_rate;
        case 'PiecewiseStepSize', opt.step_kind = 1; opt.lr = G.learning_rate; opt.decay_interval = G.decay_interval; opt.t0 = G.t;
        case 'HybridGradient',    opt.step_kind = 2; opt.lr = G.lr; opt.decay_interval = G.decay_interval; opt.hybrid_strategy = G.strategy; opt.t0 = G.t;
        otherwise, error('desc_amd:Gradient', 'params.Gradient must be a ConstantStepSize, PiecewiseStepSize or HybridGradient object');
    end
    opt.seed = 0;   if isfield(params, 'seed'),   opt.seed = params.seed;     end
    opt.device = 0; if isfield(params, 'device'), opt.device = params.device; end
    opt.verbose = 1;
    disp('compute R cycle'); disp('S0Mat'); disp('Initialization completed!'); disp('Reweighting Procedure Started ...')   % DESC.m:132,145,160,162
    [R_est, R_init, S_sorted, info] = desc_amd_mex('desc', int32(IndS - 1), double(RijMat(:,:,perm)), opt);
    disp('DONE!')                                     % DESC.m:313
    if isa(G, 'PiecewiseStepSize') || isa(G, 'HybridGradient'), G.t = info.t_end; end
    S_vec = zeros(1, size(Ind,1));
    S_vec(perm) = S_sorted;
end
