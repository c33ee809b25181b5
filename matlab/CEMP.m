%% CEMP -- drop-in replacement of the reference's Algorithms/CEMP.m:24
function SVec = CEMP(Ind, RijMat, CEMP_parameters)
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    seed = 0; if isfield(CEMP_parameters, 'seed'), seed = CEMP_parameters.seed; end
    S = desc_amd_mex('cemp', int32(IndS - 1), double(RijMat(:,:,perm)), double(CEMP_parameters.reweighting(:)'), ...
                     CEMP_parameters.max_iter, CEMP_parameters.nsample, seed);
    SVec = zeros(1, size(Ind,1)); SVec(perm) = S;
end
