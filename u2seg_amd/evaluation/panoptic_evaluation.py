"""Panoptic predictions in COCO panoptic format, with the U2Seg cluster -> category mapping applied
(detectron2/evaluation/panoptic_evaluation.py:24-190).

The evaluator looks for ./hungarian_matching/{semantic,instance}_mapping.json when it is built: without them it is in
"hungarian_matching" mode and stores the predictions as they are; with them ("eval") every segment's category goes through
the mapping - a thing cluster to the dataset id of its category, an unsupervised stuff class to cluster_num(300) + its
supercategory - and segments whose cluster has no mapping are erased from the id map.  evaluate() gathers the ranks,
writes the pngs and predictions.json and computes PQ / SQ / RQ with panopticapi's pq_compute when that package is
importable, with the restatement in evaluation/pq.py otherwise."""
import io
import itertools
import json
import os
import tempfile

from PIL import Image

from ..data.catalog import MetadataCatalog
from ..data.pseudo_panoptic import id2rgb
from .evaluator import DatasetEvaluator, gather_to_rank0

EVAL_CLUSTER_NUM = 300
MAPPING_DIR = "./hungarian_matching"


class COCOPanopticEvaluator(DatasetEvaluator):
    def __init__(self, dataset_name, output_dir=None):
        self._metadata = MetadataCatalog.get(dataset_name)
        self._thing_dataset_id = {v: k for k, v in self._metadata.thing_dataset_id_to_contiguous_id.items()}
        self._stuff_dataset_id = {i: EVAL_CLUSTER_NUM + i for i in range(1, 16)}
        self._stuff_dataset_id[0] = 0
        self._output_dir = output_dir
        if output_dir is not None:
            os.makedirs(output_dir, exist_ok=True)
        sem, ins = (os.path.join(MAPPING_DIR, n) for n in ("semantic_mapping.json", "instance_mapping.json"))
        if os.path.exists(sem):
            self.mode = "eval"
            self.semantic_mapping_dict = json.load(open(sem))
            self.instance_mapping_dict = json.load(open(ins))
        else:
            self.mode = "hungarian_matching"
        self.reset()

    def reset(self):
        self._predictions = []

    def _mapped(self, segment):
        """The segment with its dataset category id, or None when its cluster has no counterpart."""
        isthing = segment.pop("isthing", None)
        if isthing is None:
            return segment  # the model already speaks dataset ids
        table, to_dataset = ((self.instance_mapping_dict, self._thing_dataset_id) if isthing is True
                             else (self.semantic_mapping_dict, self._stuff_dataset_id))
        target = table[str(segment["category_id"])]
        if target == -1:
            return None
        segment["category_id"] = to_dataset[target]
        return segment

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            ids, segments = out["panoptic_seg"]
            ids = ids.cpu().numpy().copy()  # edited below: never the model's own output (shared with other evaluators)
            segments = [dict(seg) for seg in segments] if segments is not None else None
            assert segments is not None, "the PanopticFPN path always returns segments_info"
            if self.mode != "hungarian_matching":
                kept = []
                for seg in segments:
                    sid = seg["id"]
                    seg = self._mapped(seg)
                    if seg is None:
                        ids[ids == sid] = 0
                    else:
                        kept.append(seg)
                segments = kept
            with io.BytesIO() as buf:
                Image.fromarray(id2rgb(ids)).save(buf, format="PNG")
                png = buf.getvalue()
            name = os.path.splitext(os.path.basename(inp["file_name"]))[0] + ".png"
            self._predictions.append({"image_id": inp["image_id"], "file_name": name, "png_string": png,
                                      "segments_info": segments})

    def evaluate(self):
        parts = gather_to_rank0(self._predictions)
        if parts is None:
            return None
        predictions = list(itertools.chain(*parts))
        pred_dir = self._output_dir or tempfile.mkdtemp(prefix="panoptic_eval")
        for p in predictions:
            with open(os.path.join(pred_dir, p["file_name"]), "wb") as f:
                f.write(p.pop("png_string"))
        gt_json = self._metadata.get("panoptic_json")
        json_data = json.load(open(gt_json)) if gt_json and os.path.isfile(gt_json) else {}
        json_data["annotations"] = predictions
        predictions_json = os.path.join(pred_dir, "predictions.json")
        with open(predictions_json, "w") as f:
            f.write(json.dumps(json_data))
        result = {"predictions_json": predictions_json, "num_images": len(predictions)}
        gt_folder = self._metadata.get("panoptic_root")
        if not (gt_json and os.path.isfile(gt_json) and gt_folder and os.path.isdir(gt_folder)):
            return {"panoptic_seg": result}  # no panoptic ground truth on disk: the converted predictions are the output
        try:
            from panopticapi.evaluation import pq_compute
        except ImportError:
            from .pq import pq_compute  # the same procedure restated (parity unpinned, see evaluation/pq.py)

            result["pq_implementation"] = "u2seg_amd.evaluation.pq"
        pq = pq_compute(gt_json, predictions_json, gt_folder=gt_folder, pred_folder=pred_dir)
        for group, suffix in (("All", ""), ("Things", "_th"), ("Stuff", "_st")):
            for key in ("pq", "sq", "rq"):
                result[key.upper() + suffix] = 100 * pq[group][key]
        return {"panoptic_seg": result}
