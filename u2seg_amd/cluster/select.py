"""Density-based pick of one representative per cluster (u2seg/Instance_Clustering/shared/utils/nn_utils.py:408-439,
called from selective_labeling/usl-imagenet.py:139-152 right after k-means with neighbors_dist = d_knns.mean(1)).

The reference loops over the clusters and masks the full label vector once per cluster (K passes over N); here one stable
two-key sort puts every cluster's rows together in order of increasing distance (ties: smaller row index), and the first
row of each run is the answer - the same result in O(N log N) on whatever device the tensors live on."""
import numpy as np
import torch


def get_selection_without_reg(cluster_labels, neighbors_dist, centroid_ordering, final_sample_num):
    """For every cluster id in `centroid_ordering` (an int n means range(n)) that has members: the row with the smallest
    neighbors_dist.  Returns a numpy int array in the order of `centroid_ordering`, cut to final_sample_num; fewer
    non-empty clusters than that is an error, as in the reference."""
    labels = torch.as_tensor(cluster_labels).reshape(-1).long()
    dist = torch.as_tensor(neighbors_dist).reshape(-1).to(labels.device)
    if isinstance(centroid_ordering, int):
        centroid_ordering = range(centroid_ordering)
    ordering = torch.as_tensor(list(centroid_ordering), dtype=torch.long, device=labels.device)
    by_dist = torch.sort(dist, stable=True).indices            # rows by distance, equal distances by row index
    by_label = torch.sort(labels[by_dist], stable=True)        # ... then grouped by cluster, order inside a group kept
    rows = by_dist[by_label.indices]
    sorted_labels = by_label.values
    first = torch.ones_like(sorted_labels, dtype=torch.bool)
    first[1:] = sorted_labels[1:] != sorted_labels[:-1]
    best_label, best_row = sorted_labels[first], rows[first]
    k = int(max(int(ordering.max()) if ordering.numel() else -1, int(best_label.max()) if best_label.numel() else -1)) + 1
    table = torch.full((max(k, 1),), -1, dtype=torch.long, device=labels.device)
    table[best_label] = best_row
    picked = table[ordering]
    selected = picked[picked >= 0].cpu().numpy()
    assert selected.shape[0] >= final_sample_num, "Insufficient data: expected: {}, actual: {}".format(
        final_sample_num, selected.shape[0])
    return selected[:final_sample_num]


def cluster_label_table(names, cluster_labels):
    """{"<crop file name>": cluster id}: the form in which the clustering result enters the label preparation
    (datasets/prepare_ours/generate_classaware_instanceseg_annotations.py:38,55 reads `cluster_results[str(ins_id) + ".jpg"]`)."""
    labels = np.asarray(torch.as_tensor(cluster_labels).cpu()).tolist()
    assert len(names) == len(labels)
    return {str(n): int(c) for n, c in zip(names, labels)}
