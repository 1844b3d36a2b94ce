from .backbone import BACKBONE_REGISTRY, FPN, ResNet, build_backbone, build_resnet_backbone, build_resnet_fpn_backbone
from .panoptic_fpn import META_ARCH_REGISTRY, GeneralizedRCNN, PanopticFPN, build_model
from .roi_heads import (ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY, ROI_MASK_HEAD_REGISTRY, CascadeROIHeads,
                        FastRCNNConvFCHead, FastRCNNOutputLayers, MaskRCNNConvUpsampleHead, ROIPooler, StandardROIHeads,
                        build_box_head, build_mask_head, build_roi_heads)
from .rpn import (ANCHOR_GENERATOR_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, RPN, RPN_HEAD_REGISTRY, DefaultAnchorGenerator,
                  StandardRPNHead, build_anchor_generator, build_proposal_generator)
from .sampling import set_key_source, set_permutation_source, subsample_labels
from .semantic_seg import SEM_SEG_HEADS_REGISTRY, SemSegFPNHead, build_sem_seg_head
