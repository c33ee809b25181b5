%% DESC -- drop-in replacement of the reference's Algorithms/DESC.m:14 (called at
%% Demo/compare_algorithms.m:72):   [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
%% PGD (DESC.m:16-261) -> GCW (:263) -> reweighted LAA refinement (:265-313), all on the MI355X.
function [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
    S_vec = DESC_PGD(Ind, RijMat, params);
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    R_init = desc_amd_mex('gcw', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm));
    disp('Rotation Initialized!')                 % DESC.m:283
    disp('Start DESC refinement ...')             % DESC.m:284
    R_est = desc_amd_mex('refine', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm), R_init);
    disp('DONE!')                                 % DESC.m:313
end
