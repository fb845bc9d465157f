def fasterrcnn_resnet50_fpn(*a, **k):
    raise RuntimeError("torchvision stub")
retinanet_resnet50_fpn = retinanet_resnet50_fpn_v2 = fasterrcnn_resnet50_fpn_v2 = fasterrcnn_resnet50_fpn
class _W:
    DEFAULT = None
    COCO_V1 = None
FasterRCNN_ResNet50_FPN_Weights = RetinaNet_ResNet50_FPN_Weights = _W
RetinaNet_ResNet50_FPN_V2_Weights = FasterRCNN_ResNet50_FPN_V2_Weights = _W
