"""pybnesian_amd — MI355X (gfx950) implementation of PyBNesian's KDE/CKDE log-likelihood and
LinearGaussian/BIC/BGe/CV-likelihood scoring hot path, behind the reference's own class names.

Only the hot path of SURVEY.md §8 lives here.  Everything numerical runs in libpbn_hip.so (hand-written
HIP kernels behind the C ABI of include/pbn_hip.h); there is no CPU fallback.
"""
from ._lib import SingularCovarianceData, load as load_library  # noqa: F401
from .dataset import Context, CrossValidation, DeviceTable, HoldOut, default_context  # noqa: F401
from .factors import (CKDE, HCKDE, MLE, Assignment, CLinearGaussianCPD, DiscreteFactor, DiscreteFactorParams, Factor, LinearGaussianCPD,  # noqa: F401
                      LinearGaussianParams, MLEDiscreteFactor, MLELinearGaussianCPD)
from .kde import KDE, UCV, BandwidthSelector, NormalReferenceRule, ProductKDE, ScottsBandwidth  # noqa: F401

from .learning import (AddArc, ArcOperator, ArcOperatorSet, Callback, Operator, OperatorSet, SaveModel, ChangeNodeType, ChangeNodeTypeSet, FlipArc, GreedyHillClimbing,  # noqa: F401
                       LocalScoreCache, MMHC, OperatorPool, OperatorTabuSet, RemoveArc, hc)
from .dynamic import (DMMHC, DynamicBayesianNetwork, DynamicGaussianNetwork, DynamicKDENetwork, DynamicSemiparametricBN, DynamicBGe, DynamicBIC, DynamicChiSquare, DynamicCVLikelihood,  # noqa: F401
                      DynamicDataFrame, DynamicHoldoutLikelihood, DynamicIndependenceTestAdaptator, DynamicLinearCorrelation,
                      DynamicMutualInformation, DynamicScoreAdaptator, DynamicValidatedLikelihood)
from .independences import ChiSquare, IndependenceTest, KMutualInformation, LinearCorrelation, MutualInformation  # noqa: F401
from .models import (BayesianNetwork, BayesianNetworkType, FactorType, ConditionalBayesianNetwork, ConditionalCLGNetwork, ConditionalGaussianNetwork,  # noqa: F401
                     ConditionalKDENetwork, ConditionalSemiparametricBN, CKDEType, CLGNetwork, CLGNetworkType, DiscreteFactorType, GaussianNetwork, GaussianNetworkType, KDENetwork, KDENetworkType,  # noqa: F401
                     LinearGaussianCPDType, SemiparametricBN, SemiparametricBNType, UnknownFactorType, load,
                     ConditionalDiscreteBN, ConditionalHeterogeneousBN, ConditionalHomogeneousBN, DiscreteBN, DiscreteBNType,
                     HeterogeneousBN, HeterogeneousBNType, HomogeneousBN, HomogeneousBNType, Dag, ConditionalDag)
from .dynamic import (DynamicBDe, DynamicBayesianNetworkBase, DynamicKMutualInformation, DynamicCLGNetwork, DynamicDiscreteBN, DynamicHeterogeneousBN, DynamicHomogeneousBN,  # noqa: F401
                      DynamicIndependenceTest, DynamicScore)
from .models import BayesianNetworkBase, ConditionalBayesianNetworkBase  # noqa: F401
from .scores import (Args, Arguments, BDe, BGe, BIC, CVLikelihood, HoldoutLikelihood, Kwargs, Score, ValidatedLikelihood,  # noqa: F401
                     ValidatedScore)

__all__ = [
    "BIC", "BGe", "CVLikelihood", "HoldoutLikelihood", "ValidatedLikelihood", "GreedyHillClimbing", "hc",
    "ArcOperatorSet", "ChangeNodeTypeSet", "OperatorPool", "OperatorTabuSet", "LocalScoreCache", "AddArc", "RemoveArc", "FlipArc", "ChangeNodeType",
    "GaussianNetwork", "SemiparametricBN", "KDENetwork", "BayesianNetwork", "LinearGaussianCPDType", "CKDEType",
    "GaussianNetworkType", "SemiparametricBNType", "KDENetworkType", "CLGNetwork", "CLGNetworkType", "DiscreteFactorType",
    "KDE", "ProductKDE", "CKDE", "Factor", "LinearGaussianCPD", "MLE", "HCKDE", "CLinearGaussianCPD", "DiscreteFactor", "BandwidthSelector", "NormalReferenceRule", "ScottsBandwidth", "UCV",
    "SingularCovarianceData", "Score", "ValidatedScore", "Arguments", "Args", "Kwargs", "ConditionalBayesianNetwork", "ConditionalGaussianNetwork", "ConditionalKDENetwork", "ConditionalSemiparametricBN", "ConditionalCLGNetwork", "CrossValidation", "HoldOut", "FactorType", "BayesianNetworkType", "UnknownFactorType", "DynamicDataFrame", "DynamicBayesianNetwork", "DynamicGaussianNetwork", "DynamicSemiparametricBN", "DynamicKDENetwork", "DMMHC", "DynamicBIC", "DynamicBGe", "DynamicCVLikelihood", "DynamicHoldoutLikelihood", "DynamicValidatedLikelihood", "DynamicLinearCorrelation", "DynamicMutualInformation", "DynamicChiSquare", "DynamicScoreAdaptator", "DynamicIndependenceTestAdaptator", "Callback", "MMHC", "IndependenceTest", "LinearCorrelation", "MutualInformation", "ChiSquare", "load", "Context", "DeviceTable", "default_context", "load_library",
    "DiscreteBN", "DiscreteBNType", "HomogeneousBN", "HomogeneousBNType", "HeterogeneousBN", "HeterogeneousBNType", "ConditionalDiscreteBN",
    "ConditionalHomogeneousBN", "ConditionalHeterogeneousBN", "DynamicDiscreteBN", "DynamicCLGNetwork", "DynamicHomogeneousBN",
    "DynamicHeterogeneousBN", "Dag", "ConditionalDag", "Operator", "ArcOperator", "OperatorSet", "SaveModel", "LinearGaussianParams",
    "DiscreteFactorParams", "MLELinearGaussianCPD", "MLEDiscreteFactor", "BayesianNetworkBase", "ConditionalBayesianNetworkBase",
    "DynamicBayesianNetworkBase", "DynamicScore", "DynamicIndependenceTest", "BDe", "DynamicBDe", "KMutualInformation",
    "DynamicKMutualInformation", "Assignment",
]
